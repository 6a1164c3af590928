#!/bin/bash
# CG product of the bench matrix: image-free CSR kernel (default) against the SELL-64 image kernel, alternating processes on one box.
cd "${GRAFT_REPO_ROOT:-.}"
export PYNAMA_MATFREE=0
for i in 1 2 3; do
  python3 tools/prof_case.py cg 215 10 2>&1 | grep "cg ms" | sed "s/^/csr values:  /"
  PYNAMA_SELL_IMAGE=1 python3 tools/prof_case.py cg 215 10 2>&1 | grep "cg ms" | sed "s/^/SELL image:  /"
done
