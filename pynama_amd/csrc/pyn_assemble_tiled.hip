// Tiled owner-computes assembly for Q1 hexahedra (scalar forms): atomics-free in HBM.
//
// Reference work replaced: the per-cell loop of FreeSlip.buildKLEMats (src/cases/base_problem.py:
// 504-547) for the scalar Laplacian block of spectral.py:117-131.
//
// Design (DESIGN.md "assembly v2"):
//   * the owned rows are split into patches (<= PATCH_MAX_ROWS rows; for box meshes 7x7x7 node
//     tiles), each patch is ONE workgroup; the workgroup integrates every element that touches one
//     of its rows (one element per lane, everything in registers), accumulates the rows it owns in
//     LDS (ds_add_f64) and finally writes each CSR row exactly once with plain coalesced stores.
//     Elements on patch interfaces are integrated by every patch that needs them (no HBM atomics,
//     no zero-fill pass, bitwise-identical structure between runs up to LDS add order).
//   * a "patch plan" built once on the device after the symbolic phase gives, per (patch, element):
//     the element id, the LDS row slot of each of its 8 nodes (or "not mine") and the in-row CSR
//     slot of each of the 64 (row, col) pairs -- the classic FEM scatter map, stored SoA so that
//     one lane's 16 bytes sit next to its neighbour's.
#include <hipcub/hipcub.hpp>

#include "pyn_internal.h"
#include "pyn_q1_hex.h"

namespace {

constexpr int KLE_MAX_ROWS = 36;      // 4*3*3 node tiles: 36 rows * 27 cols * 9 * 8 B = 70 KB of LDS
constexpr int PATCH_MAX_ROWS = 352;   // 7*7*7 = 343 rows -> 74 KB of LDS accumulators at 27 cols

__device__ inline int find_slot_t(const int32_t* __restrict__ colidx, int lo, int len, int col) {
  int l = 0, h = len;
  while (l < h) {
    int m = (l + h) >> 1;
    if (colidx[lo + m] < col)
      l = m + 1;
    else
      h = m;
  }
  return l;
}

// ---- plan construction ----------------------------------------------------------------------
__global__ void plan_node_maps_kernel(const int32_t* __restrict__ p_rowptr, const int32_t* __restrict__ p_rows, int n_patch,
                                      int32_t* __restrict__ node2patch, int32_t* __restrict__ node2slot) {
  for (int p = blockIdx.x; p < n_patch; p += gridDim.x) {
    int lo = p_rowptr[p], hi = p_rowptr[p + 1];
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
      node2patch[p_rows[i]] = p;
      node2slot[p_rows[i]] = i - lo;
    }
  }
}

__global__ void plan_emit_kernel(const int32_t* __restrict__ conn, int64_t n_elem, int nn, int64_t n_owned,
                                 const int32_t* __restrict__ node2patch, unsigned long long* __restrict__ keys) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (int64_t)gridDim.x * blockDim.x) {
    int pp[8];
    int cnt = 0;
    for (int a = 0; a < nn; ++a) {
      int node = conn[e * nn + a];
      int p = node < n_owned ? node2patch[node] : -1;
      bool dup = p < 0;
      for (int j = 0; j < cnt && !dup; ++j) dup = pp[j] == p;
      if (!dup) pp[cnt++] = p;
    }
    for (int j = 0; j < nn; ++j)
      keys[e * nn + j] = j < cnt ? (((unsigned long long)pp[j] << 32) | (unsigned long long)e) : ~0ull;
  }
}

__global__ void plan_count_valid_kernel(const unsigned long long* __restrict__ keys, int64_t n, int64_t* __restrict__ out) {
  // keys sorted: first index holding ~0
  int64_t l = 0, h = n;
  while (l < h) {
    int64_t m = (l + h) >> 1;
    if (keys[m] != ~0ull)
      l = m + 1;
    else
      h = m;
  }
  *out = l;
}

__global__ void plan_fill_kernel(const unsigned long long* __restrict__ keys, int64_t npe, const int32_t* __restrict__ conn,
                                 int nn, int64_t n_owned, const int32_t* __restrict__ node2patch,
                                 const int32_t* __restrict__ node2slot, const int32_t* __restrict__ rowptr,
                                 const int32_t* __restrict__ colidx, int32_t* __restrict__ p_elem,
                                 int32_t* __restrict__ p_ecount, uint4* __restrict__ rowslot4, uint4* __restrict__ kmap4) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npe; t += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long k = keys[t];
    int p = (int)(k >> 32);
    int64_t e = (int64_t)(k & 0xffffffffull);
    p_elem[t] = (int32_t)e;
    atomicAdd(&p_ecount[p], 1);
    unsigned short rs[8];
    unsigned char km[64];
    for (int a = 0; a < 8; ++a) rs[a] = 0xFFFF;
    for (int i = 0; i < 64; ++i) km[i] = 0xFF;
    for (int a = 0; a < nn; ++a) {   // nn = 8 (hexahedra) or 4 (tetrahedra): same record, partly used
      int node = conn[e * nn + a];
      bool mine = node < n_owned && node2patch[node] == p;
      rs[a] = mine ? (unsigned short)node2slot[node] : (unsigned short)0xFFFF;
      int lo = 0, len = 0;
      if (mine) {
        lo = rowptr[node];
        len = rowptr[node + 1] - lo;
      }
      for (int b = 0; b < nn; ++b) km[a * 8 + b] = mine ? (unsigned char)find_slot_t(colidx, lo, len, conn[e * nn + b]) : 0xFF;
    }
    uint4 r;
    r.x = rs[0] | ((unsigned)rs[1] << 16);
    r.y = rs[2] | ((unsigned)rs[3] << 16);
    r.z = rs[4] | ((unsigned)rs[5] << 16);
    r.w = rs[6] | ((unsigned)rs[7] << 16);
    rowslot4[t] = r;
    for (int j = 0; j < 4; ++j) {
      uint4 v;
      const unsigned char* s = km + 16 * j;
      v.x = s[0] | (s[1] << 8) | (s[2] << 16) | ((unsigned)s[3] << 24);
      v.y = s[4] | (s[5] << 8) | (s[6] << 16) | ((unsigned)s[7] << 24);
      v.z = s[8] | (s[9] << 8) | (s[10] << 16) | ((unsigned)s[11] << 24);
      v.w = s[12] | (s[13] << 8) | (s[14] << 16) | ((unsigned)s[15] << 24);
      kmap4[(int64_t)j * npe + t] = v;
    }
  }
}

// ---- the numeric kernels ------------------------------------------------------------------------
// Store phase shared by the scalar patch kernels: write every owned row once: A gets the free columns, Arhs the
// imposed ones (negated); imposed rows become identity rows (mat_generator.py:113-118).  Half a wave per row
// (<= 32 entries), rows and per-entry bytes are fetched UNROLL at a time so that no store waits on a load it
// does not need.
__device__ __forceinline__ void plan_store_rows(const TileArgs& T, const double* acc, const int* rmeta, const unsigned* cflag,
                                                int r_lo, int nrows, int ml, int tid) {
  double* __restrict__ outA = T.A;
  double* __restrict__ outR = T.Arhs;
  const int32_t* __restrict__ colidx = T.colidx;
  const int half = tid >> 5, k = tid & 31;                  // TILE_THREADS/32 half-waves
  constexpr int NH = TILE_THREADS / 32, UNROLL = 4;
  for (int s0 = half; s0 < nrows; s0 += NH * UNROLL) {
    int lo[UNROLL], m1[UNROLL];
    unsigned char cb[UNROLL];
    double v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int slot = s0 + u * NH;
      const bool ok = slot < nrows;
      lo[u] = ok ? rmeta[2 * slot] : 0;
      m1[u] = ok ? rmeta[2 * slot + 1] : 0;
      const bool act = k < (m1[u] & 0xFFFF);
      v[u] = act ? acc[slot * ml + k] : 0.0;
      cb[u] = act ? ((cflag[slot] >> k) & 1u) : 0;
      if (act && (m1[u] >> 16)) cb[u] = (colidx[lo[u] + k] == T.p_rows[r_lo + slot]) ? 2 : 3;  // imposed row: diag?
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (k >= (m1[u] & 0xFFFF)) continue;
      double va, vr;
      if (m1[u] >> 16) {
        va = vr = (cb[u] == 2) ? 1.0 : 0.0;
      } else if (cb[u]) {
        va = 0.0;
        vr = -v[u];
      } else {
        va = v[u];
        vr = 0.0;
      }
      outA[lo[u] + k] = va;
      if (outR) outR[lo[u] + k] = vr;
    }
  }
}

template <int ABLATE>
__global__ void __launch_bounds__(TILE_THREADS, 2) assemble_q1_hex_tiled_kernel(TileArgs T) {
  extern __shared__ __align__(16) double acc[];  // [maxrows][maxlen] accumulators, then per-row meta
  const int p = blockIdx.x;
  const int r_lo = T.p_rowptr[p];
  const int nrows = T.p_rowptr[p + 1] - r_lo;
  const int e_lo = T.p_eptr[p];
  const int ne = T.p_eptr[p + 1] - e_lo;
  const int ml = T.maxlen;
  int* rmeta = reinterpret_cast<int*>(acc + (size_t)T.maxrows * ml);  // [maxrows][2]: csr offset, len | bc<<16
  unsigned* cflag = reinterpret_cast<unsigned*>(rmeta + 2 * T.maxrows);  // [maxrows]: bit k = column k imposed
  const int tid = threadIdx.x;

  // ---- issue the first dependent loads of every chain before touching LDS (latency overlap)
  int e_cur = (tid < ne) ? T.p_elem[e_lo + tid] : -1;
  int row_a = (tid < nrows) ? T.p_rows[r_lo + tid] : -1;
  int row_b = (tid + TILE_THREADS < nrows) ? T.p_rows[r_lo + tid + TILE_THREADS] : -1;
  for (int i = tid; i < nrows * ml; i += TILE_THREADS) acc[i] = 0.0;
  for (int i = tid; i < nrows; i += TILE_THREADS) cflag[i] = 0u;
  int4 c0 = make_int4(0, 0, 0, 0), c1 = c0;
  if (e_cur >= 0) {
    c0 = reinterpret_cast<const int4*>(T.conn)[(int64_t)e_cur * 2];
    c1 = reinterpret_cast<const int4*>(T.conn)[(int64_t)e_cur * 2 + 1];
  }
  if (row_a >= 0) {
    const int lo = T.rowptr[row_a], hi = T.rowptr[row_a + 1];
    rmeta[2 * tid] = lo;
    rmeta[2 * tid + 1] = (hi - lo) | ((T.bcmask && T.bcmask[row_a]) ? (1 << 16) : 0);
  }
  if (row_b >= 0) {
    const int lo = T.rowptr[row_b], hi = T.rowptr[row_b + 1];
    rmeta[2 * (tid + TILE_THREADS)] = lo;
    rmeta[2 * (tid + TILE_THREADS) + 1] = (hi - lo) | ((T.bcmask && T.bcmask[row_b]) ? (1 << 16) : 0);
  }
  __syncthreads();

  for (int base = 0; base < ne; base += TILE_THREADS) {
    const int t = base + tid;
    // prefetch the next round's element id + connectivity while this round computes
    const int tn = t + TILE_THREADS;
    const int e_nxt = (tn < ne) ? T.p_elem[e_lo + tn] : -1;
    if (e_cur >= 0) {
      const int64_t pe = (int64_t)e_lo + t;
      const int nd[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      double X[8][3];
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const double* q = T.xyz + (int64_t)nd[a] * 3;
        X[a][0] = q[0];
        X[a][1] = q[1];
        X[a][2] = q[2];
      }
      unsigned bcn = 0;  // bit b = node b imposed
      if (T.bcmask) {
#pragma unroll
        for (int b = 0; b < 8; ++b) bcn |= (T.bcmask[nd[b]] ? 1u : 0u) << b;
      }
      const uint4 rs4 = T.rowslot4[pe];
      uint4 km4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) km4[j] = T.kmap4[(int64_t)j * T.npe + pe];
      if (e_nxt >= 0) {
        c0 = reinterpret_cast<const int4*>(T.conn)[(int64_t)e_nxt * 2];
        c1 = reinterpret_cast<const int4*>(T.conn)[(int64_t)e_nxt * 2 + 1];
      }
      double L[36];
#pragma unroll
      for (int i = 0; i < 36; ++i) L[i] = 0.0;
      if (ABLATE == 2) {  // no quadrature: keep the loads live
#pragma unroll
        for (int i = 0; i < 36; ++i) L[i] = X[i % 8][i % 3];
      } else if (T.aff && __all(element_is_affine(T, X) ? 1 : 0)) {
        affine_laplace(T, X, L);  // whole wave on parallelepipeds (uniform box meshes)
      } else if (T.lean) {        // (uniform) standard tables: closed form of the same rule (pyn_q1_hex.h)
        q1_laplace_lean36(X, L);
      } else {
#pragma nounroll
        for (int g = 0; g < 8; ++g) gauss_point(T, g, X, L);
      }
      const unsigned rsw[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned kw[4] = {km4[j].x, km4[j].y, km4[j].z, km4[j].w};
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // rows a = 2j + h
          const int a = 2 * j + h;
          const unsigned slot = (rsw[j] >> (16 * h)) & 0xFFFFu;
          if (slot != 0xFFFFu) {
            double* row = acc + slot * ml;
            if (bcn) {  // rare (boundary elements): remember which columns of this row are imposed
              unsigned m = 0;
#pragma unroll
              for (int b = 0; b < 8; ++b)
                if ((bcn >> b) & 1u) m |= 1u << ((kw[2 * h + (b >> 2)] >> (8 * (b & 3))) & 0xFFu);
              atomicOr(&cflag[slot], m);
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
              const unsigned k = (kw[2 * h + (b >> 2)] >> (8 * (b & 3))) & 0xFFu;
              if (ABLATE == 1)
                asm volatile("" ::"v"(L[tri(a, b)]), "v"(k));
              else
                atomicAdd(&row[k], L[tri(a, b)]);
            }
          }
        }
      }
    } else if (e_nxt >= 0) {
      c0 = reinterpret_cast<const int4*>(T.conn)[(int64_t)e_nxt * 2];
      c1 = reinterpret_cast<const int4*>(T.conn)[(int64_t)e_nxt * 2 + 1];
    }
    e_cur = e_nxt;
  }
  __syncthreads();

  plan_store_rows(T, acc, rmeta, cflag, r_lo, nrows, ml, tid);
}

// All-parallelepiped meshes without structured topology (imported / renumbered hexahedral meshes; checked once per
// mesh on the device): four corner loads, J = S.E from the edge vectors, L_ab from the integer reference matrices
// right before its LDS adds -- the integration of the lattice kernel's lean path behind the patch plan.
__global__ void __launch_bounds__(TILE_THREADS, 4) assemble_q1_hex_tiled_affine_kernel(TileArgs T) {
  extern __shared__ __align__(16) double acc[];
  const int p = blockIdx.x;
  const int r_lo = T.p_rowptr[p];
  const int nrows = T.p_rowptr[p + 1] - r_lo;
  const int e_lo = T.p_eptr[p];
  const int ne = T.p_eptr[p + 1] - e_lo;
  const int ml = T.maxlen;
  int* rmeta = reinterpret_cast<int*>(acc + (size_t)T.maxrows * ml);
  unsigned* cflag = reinterpret_cast<unsigned*>(rmeta + 2 * T.maxrows);
  const int tid = threadIdx.x;
  for (int i = tid; i < nrows * ml; i += TILE_THREADS) acc[i] = 0.0;
  for (int sl = tid; sl < nrows; sl += TILE_THREADS) {
    cflag[sl] = 0u;
    const int row = T.p_rows[r_lo + sl];
    const int lo = T.rowptr[row], hi = T.rowptr[row + 1];
    rmeta[2 * sl] = lo;
    rmeta[2 * sl + 1] = (hi - lo) | ((T.bcmask && T.bcmask[row]) ? (1 << 16) : 0);
  }
  __syncthreads();
  const double* __restrict__ S = T.aff + 248;
  for (int t = tid; t < ne; t += TILE_THREADS) {
    const int64_t pe = (int64_t)e_lo + t;
    const int e = T.p_elem[pe];
    const int4 c0 = reinterpret_cast<const int4*>(T.conn)[(int64_t)e * 2];
    const int4 c1 = reinterpret_cast<const int4*>(T.conn)[(int64_t)e * 2 + 1];
    const int nd[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    // corners 0, 3 (+x), 1 (+y), 4 (+z) of the closure order span the parallelepiped
    const double* q0 = T.xyz + (int64_t)nd[0] * 3;
    const double* qx = T.xyz + (int64_t)nd[3] * 3;
    const double* qy = T.xyz + (int64_t)nd[1] * 3;
    const double* qz = T.xyz + (int64_t)nd[4] * 3;
    double E[3][3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      const double o = q0[x];
      E[0][x] = qx[x] - o;
      E[1][x] = qy[x] - o;
      E[2][x] = qz[x] - o;
    }
    unsigned bcn = 0;
    if (T.bcmask) {
#pragma unroll
      for (int b = 0; b < 8; ++b) bcn |= (T.bcmask[nd[b]] ? 1u : 0u) << b;
    }
    const uint4 rs4 = T.rowslot4[pe];
    uint4 km4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) km4[j] = T.kmap4[(int64_t)j * T.npe + pe];
    double J[3][3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int x = 0; x < 3; ++x) J[d][x] = fma(S[d * 3 + 2], E[2][x], fma(S[d * 3 + 1], E[1][x], S[d * 3] * E[0][x]));
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double r = 1.0 / det;
    double Ji[3][3];
    Ji[0][0] = c00 * r;
    Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
    Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
    Ji[1][0] = c01 * r;
    Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
    Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
    Ji[2][0] = c02 * r;
    Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
    Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
    double D[3][3], M2[3][2];
    {
      constexpr int RS[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int a0 = RS[u][0], a1 = RS[3 + u][0], b1 = RS[3 + u][1];
        const double qd = det * (Ji[0][a0] * Ji[0][a0] + Ji[1][a0] * Ji[1][a0] + Ji[2][a0] * Ji[2][a0]) * (1.0 / 72.0);
        const double qm = det * (Ji[0][a1] * Ji[0][b1] + Ji[1][a1] * Ji[1][b1] + Ji[2][a1] * Ji[2][b1]) * (1.0 / 72.0);
        D[u][0] = 4.0 * qd, D[u][1] = 8.0 * qd, D[u][2] = 16.0 * qd;
        M2[u][0] = 12.0 * qm, M2[u][1] = 24.0 * qm;
      }
    }
    const unsigned rsw[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned kw[4] = {km4[j].x, km4[j].y, km4[j].z, km4[j].w};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int a = 2 * j + h;
        const unsigned slot = (rsw[j] >> (16 * h)) & 0xFFFFu;
        if (slot == 0xFFFFu) continue;
        double* row = acc + slot * ml;
        if (bcn) {
          unsigned m = 0;
#pragma unroll
          for (int b = 0; b < 8; ++b)
            if ((bcn >> b) & 1u) m |= 1u << ((kw[2 * h + (b >> 2)] >> (8 * (b & 3))) & 0xFFu);
          atomicOr(&cflag[slot], m);
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const unsigned k = (kw[2 * h + (b >> 2)] >> (8 * (b & 3))) & 0xFFu;
          double v = 0.0;
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            const int n = q1_aff_int(u, a, b);
            const int an = n < 0 ? -n : n;
            if (an == 0) continue;
            const double x = u < 3 ? D[u][an == 4 ? 0 : (an == 8 ? 1 : 2)] : M2[u - 3][an == 12 ? 0 : 1];
            v = n > 0 ? v + x : v - x;
          }
          atomicAdd(&row[k], v);
        }
      }
    }
  }
  __syncthreads();
  plan_store_rows(T, acc, rmeta, cflag, r_lo, nrows, ml, tid);
}

// Linear tetrahedra through the same patch scheme (BASELINE.json configs[4], irregular indexing): patches of
// consecutive rows (compact in the Morton numbering of imported meshes), one element per lane, constant gradients
// G = J^-1 Hrs (table-driven), 16 LDS adds, rows written once -- no HBM atomics.
__global__ void __launch_bounds__(TILE_THREADS, 3) assemble_p1_tet_tiled_kernel(TileArgs T, double wsum) {
  extern __shared__ __align__(16) double acc[];
  const int p = blockIdx.x;
  const int r_lo = T.p_rowptr[p];
  const int nrows = T.p_rowptr[p + 1] - r_lo;
  const int e_lo = T.p_eptr[p];
  const int ne = T.p_eptr[p + 1] - e_lo;
  const int ml = T.maxlen;
  int* rmeta = reinterpret_cast<int*>(acc + (size_t)T.maxrows * ml);
  unsigned* cflag = reinterpret_cast<unsigned*>(rmeta + 2 * T.maxrows);
  const int tid = threadIdx.x;
  // element id -> node ids -> coordinates is a chain of three dependent loads: the first two hops run ONE element ahead (the first
  // element's before the accumulators are cleared), so that an iteration starts with its node ids in registers
  int e_nx = 0;
  int4 cn_nx = make_int4(0, 0, 0, 0);
  if (tid < ne) {
    e_nx = T.p_elem[(int64_t)e_lo + tid];
    cn_nx = reinterpret_cast<const int4*>(T.conn)[e_nx];
  }
  for (int i = tid; i < nrows * ml; i += TILE_THREADS) acc[i] = 0.0;
  for (int sl = tid; sl < nrows; sl += TILE_THREADS) {
    cflag[sl] = 0u;
    const int row = T.p_rows[r_lo + sl];
    const int lo = T.rowptr[row], hi = T.rowptr[row + 1];
    rmeta[2 * sl] = lo;
    rmeta[2 * sl + 1] = (hi - lo) | ((T.bcmask && T.bcmask[row]) ? (1 << 16) : 0);
  }
  __syncthreads();
  const double* __restrict__ hr = T.hrs;   // [3][4] reference gradients (constant over the element)
  for (int t = tid; t < ne; t += TILE_THREADS) {
    const int64_t pe = (int64_t)e_lo + t;
    const int4 cn = cn_nx;
    if (t + TILE_THREADS < ne) {
      e_nx = T.p_elem[pe + TILE_THREADS];
      cn_nx = reinterpret_cast<const int4*>(T.conn)[e_nx];
    }
    const int nd[4] = {cn.x, cn.y, cn.z, cn.w};
    double X[4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double* q = T.xyz + (int64_t)nd[a] * 3;
      X[a][0] = q[0];
      X[a][1] = q[1];
      X[a][2] = q[2];
    }
    unsigned bcn = 0;
    if (T.bcmask) {
#pragma unroll
      for (int b = 0; b < 4; ++b) bcn |= (T.bcmask[nd[b]] ? 1u : 0u) << b;
    }
    const uint4 rs4 = T.rowslot4[pe];
    const uint4 k0 = T.kmap4[pe], k1 = T.kmap4[T.npe + pe];
    const unsigned kw[4] = {k0.x, k0.z, k1.x, k1.z};        // bytes a*8 + b, b < 4
    double J[3][3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        double sacc = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) sacc = fma(hr[d * 4 + a], X[a][x], sacc);
        J[d][x] = sacc;
      }
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double r = 1.0 / det;
    double Ji[3][3];
    Ji[0][0] = c00 * r;
    Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
    Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
    Ji[1][0] = c01 * r;
    Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
    Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
    Ji[2][0] = c02 * r;
    Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
    Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
    double G[3][4];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int a = 0; a < 4; ++a) G[x][a] = fma(Ji[x][2], hr[8 + a], fma(Ji[x][1], hr[4 + a], Ji[x][0] * hr[a]));
    const double cw = wsum * det;
    const unsigned rsw[2] = {rs4.x, rs4.y};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const unsigned slot = (rsw[a >> 1] >> (16 * (a & 1))) & 0xFFFFu;
      if (slot == 0xFFFFu) continue;
      double* row = acc + slot * ml;
      if (bcn) {
        unsigned m = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if ((bcn >> b) & 1u) m |= 1u << ((kw[a] >> (8 * b)) & 0xFFu);
        atomicOr(&cflag[slot], m);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const unsigned k = (kw[a] >> (8 * b)) & 0xFFu;
        atomicAdd(&row[k], cw * (G[0][a] * G[0][b] + G[1][a] * G[1][b] + G[2][a] * G[2][b]));
      }
    }
  }
  __syncthreads();
  plan_store_rows(T, acc, rmeta, cflag, r_lo, nrows, ml, tid);
}

// =================================================================================================
// Tiled KLE assembly (3 DOF per node): K, Krhs (WHICH = 0) and Rw (WHICH = 1) of
// FreeSlip.buildKLEMats (src/cases/base_problem.py:499-552) without HBM atomics.
//
// Same scheme as the scalar kernel with 3x3 blocks: a patch owns <= KLE_MAX_ROWS nodes, its LDS holds
// acc[slot][p][k][q] (row component p, in-row column slot k, column component q) -- exactly the layout of
// the node row in the block-CSR value array, so the store phase streams 3*len contiguous doubles per
// scalar row.  Per element (one lane each):
//   K : B_ab[p][q] = d_pq (L_ab + c aw G_a.G_b) + c (ad G_pa G_qb - aw G_qa G_pb)      (spectral.py:131,152-153)
//       L = 8-point (or affine) Laplacian block, G = reduced-point gradients, c = w_r detJ_r
//   Rw: R_ab[p][k] = eps_pmk T_m[a][b] + aw c eps_kmp G_ma H_b,  T_m = sum_g c_g H_g[a] G_g[m][b]
//       (spectral.py:132,155), done as three passes over m so that only one 8x8 T_m is live
// Dirichlet routing per DOF (base_problem.py:512-547): imposed row -> identity (K, Krhs) / zero (Rw);
// imposed column -> -value into Krhs.
struct KleArgs {
  const int32_t* conn;
  const double* xyz;
  const int32_t* rowptr;
  const int32_t* colidx;
  const uint8_t* bcmask;  // per DOF (node*3 + comp), may be null
  const int32_t* p_rowptr;
  const int32_t* p_rows;
  const int32_t* p_eptr;
  const int32_t* p_elem;
  const uint4* rowslot4;
  const uint4* kmap4;
  int64_t npe;
  int maxlen, maxrows;
  const double *w, *H, *hrs, *hcoo;          // full rule
  const double *wr, *Hr, *hrsr, *hcoor;      // reduced rule (one point)
  const double* aff;
  int ablate;    // diagnostics (PYNAMA_KLE_ABLATE): 1 = no element phase, 2 = element phase without the scatter map reads
  int aff_rw;    // the closed-form int N_a d N_b table was verified: affine elements skip the Gauss loop of Rw
  int lean;      // standard 2x2x2 Gauss tables: general elements take q1_laplace_lean36 for the Laplacian part
  double alpha_d, alpha_w;
  double* K;     // WHICH 0: K     | WHICH 1: Rw
  double* Krhs;  // WHICH 0: Krhs (may be null)
};

constexpr int KLE_THREADS = 128;

// one (m, half) pass of the Rw element block: T_m[a][b] = sum_g c_g H_g[a] G_g[m][b] for the four row
// nodes a = 4*AH .. 4*AH+3, plus the reduced-point term, scattered into the LDS rows of the patch
template <int M, int AH>
__device__ __forceinline__ void rw_pass(const KleArgs& T, const double (&X)[8][3], const unsigned (&rsw)[4],
                                        const unsigned (&kmw)[16], double* acc, int rowsz, int ml) {
  double Tm[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) Tm[i] = 0.0;
#pragma nounroll
  for (int g = 0; g < 8; ++g) {
    double G[3][8];
    const double cg = T.w[g] * point_gradients(T.hcoo + g * 24, T.hrs + g * 24, X, G);
    const double* __restrict__ Hg = T.H + g * 8;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double ha = cg * Hg[4 * AH + a];
#pragma unroll
      for (int b = 0; b < 8; ++b) Tm[a * 8 + b] = fma(ha, G[M][b], Tm[a * 8 + b]);
    }
  }
  double Gr[3][8];
  const double caw = T.wr[0] * point_gradients(T.hcoor, T.hrsr, X, Gr) * T.alpha_w;
  constexpr int P1 = (M + 1) % 3, P2 = (M + 2) % 3;
  // (curl w)_p = eps_{p m k} d_m w_k, cyclic triples (0,1,2),(1,2,0),(2,0,1): (p, m, k) = (P2, M, P1) is
  // cyclic -> +T_m at (row comp P2, col comp P1), -T_m at (P1, P2); the reduced term carries eps_{k m p}
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int an = 4 * AH + a;
    const unsigned slot = (rsw[an >> 1] >> (16 * (an & 1))) & 0xFFFFu;
    if (slot == 0xFFFFu) continue;
    double* rowp = acc + (size_t)slot * rowsz;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned k = (kmw[2 * an + (b >> 2)] >> (8 * (b & 3))) & 0xFFu;
      const double tv = Tm[a * 8 + b];
      const double red = caw * Gr[M][an] * T.Hr[b];
      atomicAdd(&rowp[(P2 * ml + k) * 3 + P1], tv - red);
      atomicAdd(&rowp[(P1 * ml + k) * 3 + P2], red - tv);
    }
  }
}

template <int WHICH>
__global__ void __launch_bounds__(KLE_THREADS, 2) assemble_q1_hex_kle_tiled_kernel(KleArgs T) {
  extern __shared__ __align__(16) double acc[];  // [maxrows][3][maxlen][3]
  const int p = blockIdx.x;
  const int r_lo = T.p_rowptr[p];
  const int nrows = T.p_rowptr[p + 1] - r_lo;
  const int e_lo = T.p_eptr[p];
  const int ne = T.p_eptr[p + 1] - e_lo;
  const int ml = T.maxlen;
  const int rowsz = 9 * ml;                                                   // doubles per node row
  int* rmeta = reinterpret_cast<int*>(acc + (size_t)T.maxrows * rowsz);      // [maxrows][2]: csr offset, len | rowbc<<16
  unsigned* cflag = reinterpret_cast<unsigned*>(rmeta + 2 * T.maxrows);      // [maxrows][3]: per q, bit k = (col k, comp q) imposed
  const int tid = threadIdx.x;

  for (int i = tid; i < nrows * rowsz; i += KLE_THREADS) acc[i] = 0.0;
  for (int i = tid; i < nrows * 3; i += KLE_THREADS) cflag[i] = 0u;
  for (int sl = tid; sl < nrows; sl += KLE_THREADS) {
    const int row = T.p_rows[r_lo + sl];
    const int lo = T.rowptr[row];
    const int len = T.rowptr[row + 1] - lo;
    int rb = 0;
    if (T.bcmask) rb = (T.bcmask[row * 3] ? 1 : 0) | (T.bcmask[row * 3 + 1] ? 2 : 0) | (T.bcmask[row * 3 + 2] ? 4 : 0);
    rmeta[2 * sl] = lo;
    rmeta[2 * sl + 1] = len | (rb << 16);
  }
  __syncthreads();

  for (int base = 0; base < ne; base += KLE_THREADS) {
    const int t = base + tid;
    if (t >= ne) continue;
    const int64_t pe = (int64_t)e_lo + t;
    const int64_t e = T.p_elem[pe];
    const int4 c0 = reinterpret_cast<const int4*>(T.conn)[e * 2];
    const int4 c1 = reinterpret_cast<const int4*>(T.conn)[e * 2 + 1];
    const int nd[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    double X[8][3];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const double* q = T.xyz + (int64_t)nd[a] * 3;
      X[a][0] = q[0];
      X[a][1] = q[1];
      X[a][2] = q[2];
    }
    unsigned bcn = 0;  // bit 3*b + q = DOF (node b, comp q) imposed
    if (WHICH == 0 && T.bcmask) {
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int q = 0; q < 3; ++q) bcn |= (T.bcmask[(int64_t)nd[b] * 3 + q] ? 1u : 0u) << (3 * b + q);
    }
    const uint4 rs4 = T.rowslot4[pe];
    const unsigned rsw[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
    uint4 km4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) km4[j] = T.kmap4[(int64_t)j * T.npe + pe];
    const unsigned kmw[16] = {km4[0].x, km4[0].y, km4[0].z, km4[0].w, km4[1].x, km4[1].y, km4[1].z, km4[1].w,
                              km4[2].x, km4[2].y, km4[2].z, km4[2].w, km4[3].x, km4[3].y, km4[3].z, km4[3].w};
    // reduced (centroid) point
    if (WHICH == 0) {
      double Gr[3][8];
      const double cr = T.wr[0] * point_gradients(T.hcoor, T.hrsr, X, Gr);
      double L[36];
#pragma unroll
      for (int i = 0; i < 36; ++i) L[i] = 0.0;
      {
        TileArgs S;
        S.w = T.w;
        S.hrs = T.hrs;
        S.hcoo = T.hcoo;
        S.aff = T.aff;
        if (T.aff && __all(element_is_affine(S, X) ? 1 : 0)) {
          affine_laplace(S, X, L);
        } else if (T.lean) {
          q1_laplace_lean36(X, L);
        } else {
#pragma nounroll
          for (int g = 0; g < 8; ++g) gauss_point(S, g, X, L);
        }
      }
      const double caw = cr * T.alpha_w, cad = cr * T.alpha_d;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const unsigned slot = (rsw[a >> 1] >> (16 * (a & 1))) & 0xFFFFu;
        if (slot == 0xFFFFu) continue;
        double* rowp = acc + (size_t)slot * rowsz;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const unsigned k = (kmw[2 * a + (b >> 2)] >> (8 * (b & 3))) & 0xFFu;
          const double s_ab = Gr[0][a] * Gr[0][b] + Gr[1][a] * Gr[1][b] + Gr[2][a] * Gr[2][b];
          const double diag = L[tri(a, b)] + caw * s_ab;
          if (bcn) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
              if ((bcn >> (3 * b + q)) & 1u) atomicOr(&cflag[slot * 3 + q], 1u << k);
          }
#pragma unroll
          for (int pp = 0; pp < 3; ++pp)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              double v = cad * Gr[pp][a] * Gr[q][b] - caw * Gr[q][a] * Gr[pp][b];
              if (pp == q) v += diag;
              atomicAdd(&rowp[(pp * ml + k) * 3 + q], v);
            }
        }
      }
    } else {
      // Rw: passes over the derivative axis m (x the two halves of the row nodes, to keep one 4x8
      // block of T_m live): (p, kc) = the two other axes in both orders
      {
#pragma nounroll
        for (int it = 0; it < 6; ++it) {  // a rolled loop keeps the passes from being interleaved (registers)
          switch (it) {
            case 0: rw_pass<0, 0>(T, X, rsw, kmw, acc, rowsz, ml); break;
            case 1: rw_pass<0, 1>(T, X, rsw, kmw, acc, rowsz, ml); break;
            case 2: rw_pass<1, 0>(T, X, rsw, kmw, acc, rowsz, ml); break;
            case 3: rw_pass<1, 1>(T, X, rsw, kmw, acc, rowsz, ml); break;
            case 4: rw_pass<2, 0>(T, X, rsw, kmw, acc, rowsz, ml); break;
            default: rw_pass<2, 1>(T, X, rsw, kmw, acc, rowsz, ml); break;
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- store: every scalar row (slot, p) = 3*len contiguous doubles, written once
  {
    double* __restrict__ outA = T.K;
    double* __restrict__ outR = T.Krhs;
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int NW = KLE_THREADS / 64;
    for (int sr = wv; sr < nrows * 3; sr += NW) {
      const int slot = sr / 3, pp = sr - slot * 3;
      const int lo = rmeta[2 * slot];
      const int m1 = rmeta[2 * slot + 1];
      const int len = m1 & 0xFFFF;
      const bool rowbc = (m1 >> (16 + pp)) & 1;
      const int64_t gbase = ((int64_t)lo * 3 + (int64_t)pp * len) * 3;
      const double* src = acc + (size_t)slot * rowsz + (size_t)pp * ml * 3;
      for (int idx = lane; idx < len * 3; idx += 64) {
        const int k = idx / 3, q = idx - k * 3;
        const double v = src[idx];
        double va, vr;
        if (rowbc) {
          const bool dg = (WHICH == 0) && q == pp && T.colidx[lo + k] == T.p_rows[r_lo + slot];
          va = vr = dg ? 1.0 : 0.0;
        } else if (WHICH == 0 && ((cflag[slot * 3 + q] >> k) & 1u)) {
          va = 0.0;
          vr = -v;
        } else {
          va = v;
          vr = 0.0;
        }
        outA[gbase + idx] = va;
        if (WHICH == 0 && outR) outR[gbase + idx] = vr;
      }
    }
  }
}

// ---- all-parallelepiped meshes (checked once per mesh on the device): closed-form element blocks, FOUR waves per
// patch.  Wave w adds the node rows {2w, 2w+1} of every element (one element per lane, rounds of 64), so the
// 576 LDS adds of an element are spread over four waves and a CU holds 12 waves instead of 6 -- the general
// kernels run one wave per SIMD, which is what bounds them.  Every wave recomputes the (cheap) Jacobian.
//   K : L_ab from the integer reference matrices (q1_aff_int), centroid gradients from Ji
//   Rw: T_m[a][b] = detJ sum_d Ji[m][d] int N_a d_d N_b (q1_mix_int), no Gauss loop, no passes
constexpr int KLE_AFF_THREADS = 256;

template <int A0>
__device__ __forceinline__ void kle_affine_k_rows(const KleArgs& T, const double (&Ji)[3][3], double det, const unsigned (&rsw)[4],
                                                  const unsigned (&kmw)[16], unsigned bcn, double* acc, unsigned* cflag,
                                                  int rowsz, int ml) {
  const double cr = T.wr[0] * det;
  const double caw = cr * T.alpha_w, cad = cr * T.alpha_d;
  const double* __restrict__ hr = T.hrsr;
  double Gr[3][8];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int a = 0; a < 8; ++a) Gr[d][a] = fma(Ji[d][2], hr[16 + a], fma(Ji[d][1], hr[8 + a], Ji[d][0] * hr[a]));
  double D[3][3], M2[3][2];
  {
    constexpr int RS[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int a0 = RS[u][0], a1 = RS[3 + u][0], b1 = RS[3 + u][1];
      const double qd = det * (Ji[0][a0] * Ji[0][a0] + Ji[1][a0] * Ji[1][a0] + Ji[2][a0] * Ji[2][a0]) * (1.0 / 72.0);
      const double qm = det * (Ji[0][a1] * Ji[0][b1] + Ji[1][a1] * Ji[1][b1] + Ji[2][a1] * Ji[2][b1]) * (1.0 / 72.0);
      D[u][0] = 4.0 * qd;
      D[u][1] = 8.0 * qd;
      D[u][2] = 16.0 * qd;
      M2[u][0] = 12.0 * qm;
      M2[u][1] = 24.0 * qm;
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int a = A0 + h;
    const unsigned slot = (rsw[a >> 1] >> (16 * (a & 1))) & 0xFFFFu;
    if (slot == 0xFFFFu) continue;
    double* rowp = acc + (size_t)slot * rowsz;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned k = (kmw[2 * a + (b >> 2)] >> (8 * (b & 3))) & 0xFFu;
      double lab = 0.0;
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int n = q1_aff_int(u, a, b);
        const int an = n < 0 ? -n : n;
        if (an == 0) continue;
        const double x = u < 3 ? D[u][an == 4 ? 0 : (an == 8 ? 1 : 2)] : M2[u - 3][an == 12 ? 0 : 1];
        lab = n > 0 ? lab + x : lab - x;
      }
      const double s_ab = Gr[0][a] * Gr[0][b] + Gr[1][a] * Gr[1][b] + Gr[2][a] * Gr[2][b];
      const double diag = lab + caw * s_ab;
      if (bcn) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if ((bcn >> (3 * b + q)) & 1u) atomicOr(&cflag[slot * 3 + q], 1u << k);
      }
#pragma unroll
      for (int pp = 0; pp < 3; ++pp)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          double v = cad * Gr[pp][a] * Gr[q][b] - caw * Gr[q][a] * Gr[pp][b];
          if (pp == q) v += diag;
          atomicAdd(&rowp[(pp * ml + k) * 3 + q], v);
        }
    }
  }
}

template <int A0>
__device__ __forceinline__ void kle_affine_rw_rows(const KleArgs& T, const double (&Ji)[3][3], double det, const unsigned (&rsw)[4],
                                                   const unsigned (&kmw)[16], double* acc, int rowsz, int ml) {
  const double caw = T.wr[0] * det * T.alpha_w;
  const double* __restrict__ hr = T.hrsr;
  double D[3][3][3];   // [m][d][4|8|16] = detJ Ji[m][d] {4, 8, 16} / 72
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double x = det * Ji[m][d] * (1.0 / 72.0);
      D[m][d][0] = 4.0 * x;
      D[m][d][1] = 8.0 * x;
      D[m][d][2] = 16.0 * x;
    }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int a = A0 + h;
    const unsigned slot = (rsw[a >> 1] >> (16 * (a & 1))) & 0xFFFFu;
    if (slot == 0xFFFFu) continue;
    double* rowp = acc + (size_t)slot * rowsz;
    double gra[3];     // alpha_w c G_r[m][a] at the centroid
#pragma unroll
    for (int m = 0; m < 3; ++m) gra[m] = caw * fma(Ji[m][2], hr[16 + a], fma(Ji[m][1], hr[8 + a], Ji[m][0] * hr[a]));
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned k = (kmw[2 * a + (b >> 2)] >> (8 * (b & 3))) & 0xFFu;
      const double hb = T.Hr[b];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double tv = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const int n = q1_mix_int(d, a, b);
          const int an = n < 0 ? -n : n;
          const double x = D[m][d][an == 4 ? 0 : (an == 8 ? 1 : 2)];
          tv = n > 0 ? tv + x : tv - x;
        }
        // (curl w)_p = eps_{p m k} d_m w_k: +T_m at (row comp m+2, col comp m+1), -T_m at (m+1, m+2); the reduced
        // term carries eps_{k m p}
        const double wv = tv - gra[m] * hb;
        const int P1 = (m + 1) % 3, P2 = (m + 2) % 3;
        atomicAdd(&rowp[(P2 * ml + k) * 3 + P1], wv);
        atomicAdd(&rowp[(P1 * ml + k) * 3 + P2], -wv);
      }
    }
  }
}

template <bool RW>
__global__ void __launch_bounds__(KLE_AFF_THREADS, 3) assemble_q1_hex_kle_affine_kernel(KleArgs T) {
  extern __shared__ __align__(16) double acc[];  // [maxrows][3][maxlen][3]
  const int p = blockIdx.x;
  const int r_lo = T.p_rowptr[p];
  const int nrows = T.p_rowptr[p + 1] - r_lo;
  const int e_lo = T.p_eptr[p];
  const int ne = T.p_eptr[p + 1] - e_lo;
  const int ml = T.maxlen;
  const int rowsz = 9 * ml;
  int* rmeta = reinterpret_cast<int*>(acc + (size_t)T.maxrows * rowsz);
  unsigned* cflag = reinterpret_cast<unsigned*>(rmeta + 2 * T.maxrows);
  const int tid = threadIdx.x, lane = tid & 63, part = tid >> 6;

  for (int i = tid; i < nrows * rowsz; i += KLE_AFF_THREADS) acc[i] = 0.0;
  for (int i = tid; i < nrows * 3; i += KLE_AFF_THREADS) cflag[i] = 0u;
  for (int sl = tid; sl < nrows; sl += KLE_AFF_THREADS) {
    const int row = T.p_rows[r_lo + sl];
    const int lo = T.rowptr[row];
    const int len = T.rowptr[row + 1] - lo;
    int rb = 0;
    if (T.bcmask) rb = (T.bcmask[row * 3] ? 1 : 0) | (T.bcmask[row * 3 + 1] ? 2 : 0) | (T.bcmask[row * 3 + 2] ? 4 : 0);
    rmeta[2 * sl] = lo;
    rmeta[2 * sl + 1] = len | (rb << 16);
  }
  __syncthreads();

  for (int base = 0; base < ne && T.ablate != 1; base += 64) {
    const int t = base + lane;
    if (t >= ne) continue;
    const int64_t pe = (int64_t)e_lo + t;
    const int64_t e = T.p_elem[pe];
    const int4 c0 = reinterpret_cast<const int4*>(T.conn)[e * 2];
    const int4 c1 = reinterpret_cast<const int4*>(T.conn)[e * 2 + 1];
    const int nd[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    double X[8][3];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const double* q = T.xyz + (int64_t)nd[a] * 3;
      X[a][0] = q[0];
      X[a][1] = q[1];
      X[a][2] = q[2];
    }
    unsigned bcn = 0;  // bit 3*b + q = DOF (node b, comp q) imposed
    if (!RW && T.bcmask) {
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int q = 0; q < 3; ++q) bcn |= (T.bcmask[(int64_t)nd[b] * 3 + q] ? 1u : 0u) << (3 * b + q);
    }
    const uint4 rs4 = T.rowslot4[pe];
    const unsigned rsw[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
    // only this wave's two rows of the scatter map
    const uint4 km = T.kmap4[(int64_t)part * T.npe + pe];
    unsigned kmw[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) kmw[j] = 0;
    double Ji[3][3];
    const double det = jacobian_inverse(T.hcoor, X, Ji);
    switch (part) {   // wave-uniform
      case 0:
        kmw[0] = km.x, kmw[1] = km.y, kmw[2] = km.z, kmw[3] = km.w;
        if (RW) kle_affine_rw_rows<0>(T, Ji, det, rsw, kmw, acc, rowsz, ml);
        else kle_affine_k_rows<0>(T, Ji, det, rsw, kmw, bcn, acc, cflag, rowsz, ml);
        break;
      case 1:
        kmw[4] = km.x, kmw[5] = km.y, kmw[6] = km.z, kmw[7] = km.w;
        if (RW) kle_affine_rw_rows<2>(T, Ji, det, rsw, kmw, acc, rowsz, ml);
        else kle_affine_k_rows<2>(T, Ji, det, rsw, kmw, bcn, acc, cflag, rowsz, ml);
        break;
      case 2:
        kmw[8] = km.x, kmw[9] = km.y, kmw[10] = km.z, kmw[11] = km.w;
        if (RW) kle_affine_rw_rows<4>(T, Ji, det, rsw, kmw, acc, rowsz, ml);
        else kle_affine_k_rows<4>(T, Ji, det, rsw, kmw, bcn, acc, cflag, rowsz, ml);
        break;
      default:
        kmw[12] = km.x, kmw[13] = km.y, kmw[14] = km.z, kmw[15] = km.w;
        if (RW) kle_affine_rw_rows<6>(T, Ji, det, rsw, kmw, acc, rowsz, ml);
        else kle_affine_k_rows<6>(T, Ji, det, rsw, kmw, bcn, acc, cflag, rowsz, ml);
        break;
    }
  }
  __syncthreads();

  // ---- store: every scalar row (slot, p) = 3*len contiguous doubles, written once (as in the general kernel)
  {
    double* __restrict__ outA = T.K;
    double* __restrict__ outR = T.Krhs;
    constexpr int NW = KLE_AFF_THREADS / 64;
    for (int sr = part; sr < nrows * 3; sr += NW) {
      const int slot = sr / 3, pp = sr - slot * 3;
      const int lo = rmeta[2 * slot];
      const int m1 = rmeta[2 * slot + 1];
      const int len = m1 & 0xFFFF;
      const bool rowbc = (m1 >> (16 + pp)) & 1;
      const int64_t gbase = ((int64_t)lo * 3 + (int64_t)pp * len) * 3;
      const double* src = acc + (size_t)slot * rowsz + (size_t)pp * ml * 3;
      for (int idx = lane; idx < len * 3; idx += 64) {
        const int k = idx / 3, q = idx - k * 3;
        const double v = src[idx];
        double va, vr;
        if (rowbc) {
          const bool dg = !RW && q == pp && T.colidx[lo + k] == T.p_rows[r_lo + slot];
          va = vr = dg ? 1.0 : 0.0;
        } else if (!RW && ((cflag[slot * 3 + q] >> k) & 1u)) {
          va = 0.0;
          vr = -v;
        } else {
          va = v;
          vr = 0.0;
        }
        outA[gbase + idx] = va;
        if (!RW && outR) outR[gbase + idx] = vr;
      }
    }
  }
}

static size_t kle_lds_bytes(int max_rows, int maxlen) {
  return (size_t)max_rows * 9 * maxlen * sizeof(double) + (size_t)max_rows * 2 * sizeof(int) + (size_t)max_rows * 3 * sizeof(unsigned);
}

}  // namespace

bool pyn_q1_mixed_tables_standard(const double* w, const double* H, const double* Hrs) {  // full rule, 8 points
  for (int d = 0; d < 3; ++d)
    for (int a = 0; a < 8; ++a)
      for (int b = 0; b < 8; ++b) {
        double u = 0.0;
        for (int g = 0; g < 8; ++g) u += w[g] * H[g * 8 + a] * Hrs[g * 24 + d * 8 + b];
        if (fabs(u - q1_mix_int(d, a, b) / 72.0) > 1e-13) return false;
      }
  return true;
}

bool pyn_q1_affine_tables_standard(const double* aff) {  // aff[6][36] as built by pyn_elem_tables_set
  for (int t = 0; t < 6; ++t) {
    int idx = 0;
    for (int a = 0; a < 8; ++a)
      for (int b = a; b < 8; ++b, ++idx)
        if (fabs(aff[t * 36 + idx] - q1_aff_int(t, a, b) / 72.0) > 1e-13) return false;
  }
  return true;
}


// -------------------------------------------------------------------------------------------------
extern "C" int pyn_patch_plan_set(pyn_ctx* c, int n_patch, const int32_t* patch_ptr, const int32_t* patch_rows) {
  return pyn_patch_plan_set_kind(c, 0, n_patch, patch_ptr, patch_rows);
}

extern "C" int pyn_patch_plan_info(pyn_ctx* c, int kind, int64_t* info) {
  PYN_CHECK(c && info, "NULL argument");
  PYN_CHECK(kind == 0 || kind == 1, "plan kind must be 0 (scalar) or 1 (KLE)");
  const PatchPlan& P = c->plan[kind];
  info[0] = P.npatch;
  info[1] = P.maxrows;
  info[2] = P.maxlen;
  info[3] = P.npe;
  return PYN_OK;
}

static bool g_default_plan = false;  // pyn_patch_plan_set_kind called by ensure_default_plan

extern "C" int pyn_patch_plan_set_kind(pyn_ctx* c, int kind, int n_patch, const int32_t* patch_ptr, const int32_t* patch_rows) {
  PYN_CHECK(c, "ctx is NULL");
  PYN_CHECK(kind == 0 || kind == 1, "plan kind must be 0 (scalar) or 1 (KLE)");
  PatchPlan& P = c->plan[kind];
  const int max_rows_allowed = kind == 0 ? PATCH_MAX_ROWS : KLE_MAX_ROWS;
  PYN_HIP(hipSetDevice(c->device));
  // drop an existing plan
  (void)hipFree(P.rowptr);
  (void)hipFree(P.rows);
  (void)hipFree(P.eptr);
  (void)hipFree(P.elem);
  (void)hipFree(P.rowslot4);
  (void)hipFree(P.kmap4);
  P = PatchPlan();
  if (n_patch == 0) return PYN_OK;
  PYN_CHECK(patch_ptr && patch_rows, "NULL argument");
  PYN_CHECK(c->d_rowptr, "pyn_csr_symbolic first");
  PYN_CHECK(c->dim == 3 && (c->nn == 8 || (c->nn == 4 && kind == 0)), "patch plans are implemented for Q1 hexahedra (and, scalar forms, linear tetrahedra)");
  PYN_CHECK(patch_ptr[0] == 0 && patch_ptr[n_patch] == c->n_owned, "patches must cover the owned rows exactly once");
  int max_rows = 0;
  for (int p = 0; p < n_patch; ++p) {
    PYN_CHECK(patch_ptr[p + 1] >= patch_ptr[p], "patch_ptr not monotone");
    max_rows = std::max(max_rows, patch_ptr[p + 1] - patch_ptr[p]);
  }
  PYN_CHECK(max_rows <= max_rows_allowed, "a patch has %d rows (max %d for this kind)", max_rows, max_rows_allowed);
  {
    std::vector<uint8_t> seen((size_t)c->n_owned, 0);
    for (int64_t i = 0; i < c->n_owned; ++i) {
      PYN_CHECK(patch_rows[i] >= 0 && patch_rows[i] < c->n_owned && !seen[patch_rows[i]], "patch_rows is not a permutation of the owned rows");
      seen[patch_rows[i]] = 1;
    }
  }
  hipStream_t s = c->stream;
  PYN_HIP(hipMalloc((void**)&P.rowptr, (n_patch + 1) * sizeof(int32_t)));
  PYN_HIP(hipMalloc((void**)&P.rows, c->n_owned * sizeof(int32_t)));
  PYN_HIP(hipMemcpyAsync(P.rowptr, patch_ptr, (n_patch + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
  PYN_HIP(hipMemcpyAsync(P.rows, patch_rows, c->n_owned * sizeof(int32_t), hipMemcpyHostToDevice, s));
  DevTmp t_n2p, t_n2s, t_k0, t_k1, t_npe, t_tmp, t_cnt;  // scratch, released on every exit path
  PYN_HIP(t_n2p.alloc(c->n_owned * sizeof(int32_t)));
  PYN_HIP(t_n2s.alloc(c->n_owned * sizeof(int32_t)));
  int32_t *node2patch = t_n2p.as<int32_t>(), *node2slot = t_n2s.as<int32_t>();
  plan_node_maps_kernel<<<std::min(n_patch, 65536), 256, 0, s>>>(P.rowptr, P.rows, n_patch, node2patch, node2slot);
  const int64_t nk = c->n_elem * c->nn;
  PYN_HIP(t_k0.alloc(nk * sizeof(unsigned long long)));
  PYN_HIP(t_k1.alloc(nk * sizeof(unsigned long long)));
  PYN_HIP(t_npe.alloc(sizeof(int64_t)));
  unsigned long long *k0 = t_k0.as<unsigned long long>(), *k1 = t_k1.as<unsigned long long>();
  int64_t* d_npe = t_npe.as<int64_t>();
  int grid = (int)std::min<int64_t>((c->n_elem + 255) / 256, 65536);
  plan_emit_kernel<<<grid, 256, 0, s>>>(c->d_conn, c->n_elem, c->nn, c->n_owned, node2patch, k0);
  size_t tb = 0;
  PYN_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, k0, k1, nk, 0, 64, s));
  PYN_HIP(t_tmp.alloc(tb));
  PYN_HIP(hipcub::DeviceRadixSort::SortKeys(t_tmp.p, tb, k0, k1, nk, 0, 64, s));
  plan_count_valid_kernel<<<1, 1, 0, s>>>(k1, nk, d_npe);
  int64_t npe = 0;
  PYN_HIP(hipMemcpyAsync(&npe, d_npe, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipStreamSynchronize(s));
  PYN_CHECK(npe > 0 && npe < (int64_t)INT32_MAX, "bad patch-element count %lld", (long long)npe);
  PYN_HIP(t_cnt.alloc((n_patch + 1) * sizeof(int32_t)));
  int32_t* ecount = t_cnt.as<int32_t>();
  PYN_HIP(hipMemsetAsync(ecount, 0, (n_patch + 1) * sizeof(int32_t), s));
  PYN_HIP(hipMalloc((void**)&P.eptr, (n_patch + 1) * sizeof(int32_t)));
  PYN_HIP(hipMalloc((void**)&P.elem, npe * sizeof(int32_t)));
  PYN_HIP(hipMalloc((void**)&P.rowslot4, npe * sizeof(uint4)));
  PYN_HIP(hipMalloc((void**)&P.kmap4, 4 * npe * sizeof(uint4)));
  grid = (int)std::min<int64_t>((npe + 255) / 256, 65536);
  plan_fill_kernel<<<grid, 256, 0, s>>>(k1, npe, c->d_conn, c->nn, c->n_owned, node2patch, node2slot, c->d_rowptr, c->d_colidx,
                                        P.elem, ecount, (uint4*)P.rowslot4, (uint4*)P.kmap4);
  tb = 0;
  PYN_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, ecount, P.eptr, n_patch + 1, s));
  PYN_HIP(hipStreamSynchronize(s));
  PYN_HIP(t_tmp.alloc(tb));
  PYN_HIP(hipcub::DeviceScan::ExclusiveSum(t_tmp.p, tb, ecount, P.eptr, n_patch + 1, s));
  // max row length of the graph (LDS row stride)
  std::vector<int32_t> rp((size_t)c->n_owned + 1);
  PYN_HIP(hipMemcpyAsync(rp.data(), c->d_rowptr, (c->n_owned + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipStreamSynchronize(s));
  int maxlen = 0;
  for (int64_t i = 0; i < c->n_owned; ++i) maxlen = std::max(maxlen, rp[i + 1] - rp[i]);
  P.npatch = n_patch;
  P.user = !g_default_plan;
  P.npe = npe;
  P.maxrows = max_rows;
  P.maxlen = maxlen;
  PYN_CHECK(maxlen <= 32, "rows of %d entries: the tiled kernels handle <= 32 (Q1 hex has 27)", maxlen);
  if (kind == 0) {
    size_t lds = (size_t)max_rows * maxlen * sizeof(double) + (size_t)max_rows * 3 * sizeof(int);
    PYN_CHECK(lds <= 160 * 1024, "patch accumulators need %zu B of LDS", lds);
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_tiled_kernel<0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_tiled_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_tiled_kernel<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_p1_tet_tiled_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_tiled_affine_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  } else {
    size_t lds = kle_lds_bytes(max_rows, maxlen);
    PYN_CHECK(lds <= 160 * 1024, "KLE patch accumulators need %zu B of LDS", lds);
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_kle_tiled_kernel<0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_kle_tiled_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_kle_affine_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_kle_affine_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  return PYN_OK;
}

namespace {
// one-off check for any Q1 hex mesh (connectivity-driven): is every element a parallelepiped?
__global__ void mesh_all_affine_kernel(const int32_t* __restrict__ conn, const double* __restrict__ xyz, int64_t n_elem,
                                       TileArgs q, int* flag) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_elem) return;
  double X[8][3];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const double* p = xyz + (int64_t)conn[e * 8 + a] * 3;
    X[a][0] = p[0];
    X[a][1] = p[1];
    X[a][2] = p[2];
  }
  if (!element_is_affine(q, X)) *flag = 0;
}


}  // namespace

// 1 iff every element of the (Q1 hex) mesh is a parallelepiped; computed once per mesh, cached in the context
int pyn_mesh_all_affine(pyn_ctx* c, int* out) {
  if (!c->d_aff || c->dim != 3 || c->nn != 8) {  // tables not uploaded (yet): nothing to cache
    *out = 0;
    return PYN_OK;
  }
  if (c->mesh_affine < 0) {
    c->mesh_affine = 0;
    {
      DevTmp flag;
      PYN_HIP(flag.alloc(sizeof(int)));
      const int one = 1;
      PYN_HIP(hipMemcpyAsync(flag.p, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
      TileArgs q = TileArgs();
      q.aff = c->d_aff;
      mesh_all_affine_kernel<<<(int)((c->n_elem + 255) / 256), 256, 0, c->stream>>>(c->d_conn, c->d_xyz, c->n_elem, q, flag.as<int>());
      int h = 0;
      PYN_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      PYN_HIP(hipStreamSynchronize(c->stream));
      c->mesh_affine = h;
    }
  }
  *out = c->mesh_affine;
  return PYN_OK;
}

static int assemble_kle_tiled(pyn_ctx* c, double alpha_d, double alpha_w, double* K, double* Krhs, double* Rw, bool* handled) {
  PatchPlan& P = c->plan[1];
  if (!P.npatch || c->dim != 3 || c->nn != 8 || c->quad[0].ngp != 8 || c->quad[1].ngp != 1) return PYN_OK;
  KleArgs T;
  T.conn = c->d_conn;
  T.xyz = c->d_xyz;
  T.rowptr = c->d_rowptr;
  T.colidx = c->d_colidx;
  T.bcmask = c->d_bcmask;
  T.p_rowptr = P.rowptr;
  T.p_rows = P.rows;
  T.p_eptr = P.eptr;
  T.p_elem = P.elem;
  T.rowslot4 = (const uint4*)P.rowslot4;
  T.kmap4 = (const uint4*)P.kmap4;
  T.npe = P.npe;
  T.maxlen = P.maxlen;
  T.maxrows = P.maxrows;
  T.w = c->quad[0].w;
  T.H = c->quad[0].H;
  T.hrs = c->quad[0].Hrs;
  T.hcoo = c->quad[0].HrsCoo;
  T.wr = c->quad[1].w;
  T.Hr = c->quad[1].H;
  T.hrsr = c->quad[1].Hrs;
  T.hcoor = c->quad[1].HrsCoo;
  T.aff = getenv("PYNAMA_NO_AFFINE") ? nullptr : c->d_aff;
  T.aff_rw = (T.aff && c->aff_rw_standard) ? 1 : 0;
  T.lean = c->q1_gauss_standard && !getenv("PYNAMA_NO_LEAN") ? 1 : 0;
  {
    const char* ab = getenv("PYNAMA_KLE_ABLATE");
    T.ablate = ab ? atoi(ab) : 0;
  }
  T.alpha_d = alpha_d;
  T.alpha_w = alpha_w;
  const size_t lds = kle_lds_bytes(P.maxrows, P.maxlen);
  if (K) {
    T.K = K;
    T.Krhs = Krhs;
    int all_aff = 0;
    if (T.aff && c->aff_standard) PYN_TRY(pyn_mesh_all_affine(c, &all_aff));
    if (all_aff)
      assemble_q1_hex_kle_affine_kernel<false><<<P.npatch, KLE_AFF_THREADS, lds, c->stream>>>(T);
    else
      assemble_q1_hex_kle_tiled_kernel<0><<<P.npatch, KLE_THREADS, lds, c->stream>>>(T);
  }
  if (Rw) {
    T.K = Rw;
    T.Krhs = nullptr;
    int all_aff = 0;
    if (T.aff_rw) PYN_TRY(pyn_mesh_all_affine(c, &all_aff));
    if (all_aff)
      assemble_q1_hex_kle_affine_kernel<true><<<P.npatch, KLE_AFF_THREADS, lds, c->stream>>>(T);
    else
      assemble_q1_hex_kle_tiled_kernel<1><<<P.npatch, KLE_THREADS, lds, c->stream>>>(T);
  }
  PYN_HIP(hipGetLastError());
  *handled = true;
  return PYN_OK;
}

// Meshes without a caller-supplied plan get patches of consecutive rows: optimal for no numbering in
// particular, but any partition is valid and even 8x redundant integration beats the HBM-atomic scatter.

static int ensure_default_plan(pyn_ctx* c, int kind) {
  const bool tets = kind == 0 && c->nn == 4 && c->quad[0].const_grad;
  if (c->plan[kind].npatch || c->plan_unfit[kind] || c->dim != 3 || !(c->nn == 8 || tets) || getenv("PYNAMA_NO_AUTO_PLAN")) return PYN_OK;
  const int64_t n = c->n_owned;
  std::vector<int32_t> ptr, rows((size_t)n);
  if (c->lat.valid) {
    // structured topology: node tiles (7x7x7 for the scalar kernel, 3x3x3 for the 3x3-block kernels)
    const int t = kind == 0 ? 7 : 3;
    const Lattice& L = c->lat;
    const int64_t nxny = (int64_t)L.nx * L.ny;
    int64_t pos = 0;
    ptr.push_back(0);
    for (int z0 = 0; z0 < L.n_own; z0 += t)
      for (int y0 = 0; y0 < L.ny; y0 += t)
        for (int x0 = 0; x0 < L.nx; x0 += t) {
          for (int z = z0; z < std::min(z0 + t, L.n_own); ++z)
            for (int y = y0; y < std::min(y0 + t, L.ny); ++y)
              for (int x = x0; x < std::min(x0 + t, L.nx); ++x) rows[pos++] = (int32_t)(z * nxny + (int64_t)y * L.nx + x);
          ptr.push_back((int32_t)pos);
        }
  } else {
    // any other numbering: consecutive-row chunks
    const int chunk = kind == 0 ? 343 : 27;
    const int np = (int)((n + chunk - 1) / chunk);
    ptr.resize((size_t)np + 1);
    for (int p = 0; p <= np; ++p) ptr[p] = (int32_t)std::min<int64_t>((int64_t)p * chunk, n);
    for (int64_t i = 0; i < n; ++i) rows[i] = (int32_t)i;
  }
  g_default_plan = true;
  int rc = pyn_patch_plan_set_kind(c, kind, (int)ptr.size() - 1, ptr.data(), rows.data());
  g_default_plan = false;
  if (rc != PYN_OK && tets) {   // e.g. rows longer than the 32 entries the store phase handles: atomics kernel instead
    (void)pyn_patch_plan_set_kind(c, kind, 0, nullptr, nullptr);
    c->plan_unfit[kind] = true;
    rc = PYN_OK;
  }
  return rc;
}

int pyn_assemble_q1_tiled(pyn_ctx* c, int form, double alpha_d, double alpha_w, double* K, double* Krhs, double* Rw, double* Rd, bool* handled) {
  *handled = false;
  if (c->ho3.valid && !Rd && ((form == PYN_FORM_LAPLACE && K && !Rw) || (form == PYN_FORM_KLE && (K || (Rw && !Krhs))))) {
    // second-order (ngl = 3) structured meshes: row-run kernels without atomics (pyn_assemble_ho3.hip)
    PYN_TRY(pyn_assemble_ho3_lattice(c, form, alpha_d, alpha_w, K, Krhs, Rw, handled));
    if (*handled) return PYN_OK;
    PYN_CHECK(!getenv("PYNAMA_HO3_REQUIRE"), "PYNAMA_HO3_REQUIRE: the ngl = 3 lattice kernels declined this assembly (non-affine cell or tables missing)");
  }
  // Kernel families that cannot address a COMPACT imposed-column target (c->asm_rcrow) assemble K (and Rw) alone; run_assembly then
  // fills the compact Krhs from the elements that hold an imposed node.  Native: the ngl = 3 row-run kernels above, the KLE lattice kernels.
  double* const KrhsN = c->asm_rcrow ? nullptr : Krhs;
  const bool pend = Krhs && c->asm_rcrow;
  if (form == PYN_FORM_LAPLACE && K && !Rw && !Rd && c->lat.valid && !c->plan[0].user) {
    PYN_TRY(pyn_assemble_lattice(c, K, KrhsN, handled));
    if (*handled) {
      c->asm_krhs_pending = pend;
      return PYN_OK;
    }
  }
  if (form == PYN_FORM_KLE && (K || (Rw && !Krhs)) && !Rd && !c->plan[1].user) {   // (Rw alone is a legal request of the ABI)
    PYN_TRY(pyn_assemble_kle_lattice(c, alpha_d, alpha_w, K, Krhs, Rw, handled));
    if (*handled) return PYN_OK;
  }
  if (form == PYN_FORM_KLE && K && !Rd) PYN_TRY(ensure_default_plan(c, 1));
  if (form == PYN_FORM_LAPLACE && K && !Rw && !Rd) PYN_TRY(ensure_default_plan(c, 0));
  if (form == PYN_FORM_KLE && K && !Rd) {
    PYN_TRY(assemble_kle_tiled(c, alpha_d, alpha_w, K, KrhsN, Rw, handled));
    if (*handled) c->asm_krhs_pending = pend;
    return PYN_OK;
  }
  PatchPlan& P = c->plan[0];
  if (!P.npatch || form != PYN_FORM_LAPLACE || !K || Rw || Rd) return PYN_OK;
  const bool tets = c->dim == 3 && c->nn == 4 && c->quad[0].const_grad && !getenv("PYNAMA_NO_P1_TILED");
  if (!tets && (c->dim != 3 || c->nn != 8 || c->quad[0].ngp != 8)) return PYN_OK;
  TileArgs T;
  T.conn = c->d_conn;
  T.xyz = c->d_xyz;
  T.rowptr = c->d_rowptr;
  T.colidx = c->d_colidx;
  T.bcmask = c->d_bcmask;
  T.colbc = nullptr;
  T.p_rowptr = P.rowptr;
  T.p_rows = P.rows;
  T.p_eptr = P.eptr;
  T.p_elem = P.elem;
  T.rowslot4 = (const uint4*)P.rowslot4;
  T.kmap4 = (const uint4*)P.kmap4;
  T.npe = P.npe;
  T.n_patch = P.npatch;
  T.maxlen = P.maxlen;
  T.maxrows = P.maxrows;
  T.w = c->quad[0].w;
  T.hrs = c->quad[0].Hrs;
  T.hcoo = c->quad[0].HrsCoo;
  T.aff = getenv("PYNAMA_NO_AFFINE") ? nullptr : c->d_aff;
  T.lean = c->q1_gauss_standard && !getenv("PYNAMA_NO_LEAN") ? 1 : 0;
  T.A = K;
  T.Arhs = KrhsN;
  c->asm_krhs_pending = pend;
  size_t lds = (size_t)P.maxrows * P.maxlen * sizeof(double) + (size_t)P.maxrows * 3 * sizeof(int);
  const char* ab = getenv("PYNAMA_TILED_ABLATE");  // diagnostics only: 1 = no LDS adds, 2 = no quadrature
  const int abl = ab ? atoi(ab) : 0;
  int all_aff = 0;
  if (!tets && T.aff && c->aff_standard && !abl && !getenv("PYNAMA_NO_LEAN_PLAN")) PYN_TRY(pyn_mesh_all_affine(c, &all_aff));
  if (tets)
    assemble_p1_tet_tiled_kernel<<<P.npatch, TILE_THREADS, lds, c->stream>>>(T, c->quad[0].wsum);
  else if (all_aff)
    assemble_q1_hex_tiled_affine_kernel<<<P.npatch, TILE_THREADS, lds, c->stream>>>(T);
  else if (abl == 1)
    assemble_q1_hex_tiled_kernel<1><<<P.npatch, TILE_THREADS, lds, c->stream>>>(T);
  else if (abl == 2)
    assemble_q1_hex_tiled_kernel<2><<<P.npatch, TILE_THREADS, lds, c->stream>>>(T);
  else
    assemble_q1_hex_tiled_kernel<0><<<P.npatch, TILE_THREADS, lds, c->stream>>>(T);
  PYN_HIP(hipGetLastError());
  *handled = true;
  return PYN_OK;
}

