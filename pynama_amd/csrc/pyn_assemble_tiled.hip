// Tiled owner-computes Q1 assembly (atomics-free).  Placeholder: not built yet -> the generic
// wave-per-element kernel of pyn_assemble.hip handles every mesh.
#include "pyn_internal.h"

int pyn_assemble_q1_tiled(pyn_ctx*, int, double, double, double*, double*, double*, double*, bool* handled) {
  *handled = false;
  return PYN_OK;
}
