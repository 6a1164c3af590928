for r in 2 4 8; do echo "3D R=$r"; PYNAMA_HO3_RUN=$r python tools/ho3_case.py 3 64 3 2>&1 | grep -E "assembly|Krhs"; done
for r in 16 32 64; do echo "2D R=$r"; PYNAMA_HO3_RUN=$r python tools/ho3_case.py 2 1024 3 2>&1 | grep -E "assembly|Krhs"; done
