// General-geometry (quadrature-path) assembly of Q1 hexahedral meshes with STRUCTURED topology: z-marching kernel.
//
// The reference integrates every cell with the 2x2x2 Gauss rule (src/elements/spectral.py:117-131, driven per cell by
// src/cases/base_problem.py:505-506).  On general (non-parallelepiped) geometry that loop is FP64-VALU-bound on MI355X
// (1,770 FP64 instructions per element = 7,080 SIMD cycles per wave of elements, against 387 B of compulsory traffic), so
// this path is organised around the arithmetic:
//   * a workgroup owns a TX x TY column of rows and MARCHES through a chunk of z-planes, one element per lane and layer:
//     (TX+1)(TY+1)/(TX TY) of the elements are integrated (1.31 for 7 x 7, 1.14 for 15 x 15) instead of the 1.49 of the
//     7^3 tiles of assemble_q1_hex_lattice_kernel; the four upper corners of a layer are the lower corners of the next
//     one (coordinates stay in registers, the next plane is prefetched);
//   * the element matrix comes from the lean closed form q1_laplace_lean (pyn_q1_hex.h): Jacobian rows from the Haar
//     coefficients of the corner coordinates, gradients from the cofactor matrix (no inverse, one reciprocal per point),
//     the 28 off-diagonal entries only -- the rows of a Laplacian element matrix sum to zero, so the diagonal follows;
//   * two plane buffers in LDS (27-point stencil rows, as in the tile kernels): layer l adds into planes l and l+1, then
//     plane l is complete and leaves as x-line copies (lat_store_plain) or through the CSR-slot decode (lat_store); loads
//     are retired BEFORE the stores of a layer are issued (one in-order vmcnt queue per wave), flag loads
//     are branch-free and only consumed after the arithmetic.
// Default shape: 7 x 7 rows = 8 x 8 elements = ONE wave per workgroup (no cross-wave barrier waits, 7 workgroups per
// CU hide each other's LDS / store phases).  What was measured on the way (15 x 15 columns, a row-owning variant
// without LDS atomics, sliced copy-out, two workgroups per CU) is in DESIGN.md 5.
// Shares the lattice descriptor, meta data and store phases with pyn_assemble_lattice.hip (pyn_lattice.h).
#include "pyn_internal.h"
#include "pyn_lattice.h"
#include "pyn_q1_hex.h"

namespace {

template <int TX, int TY>
struct MarchTile {
  static constexpr int EX = TX + 1, EY = TY + 1, NT = EX * EY, NR = TX * TY, ACC = NR * 27;
  using L1 = LatTile<TX, TY, 1>;
  static constexpr size_t BYTES = 2 * (size_t)ACC * sizeof(double) + (L1::META_INTS + 2) * sizeof(int);   // + anyflag[2]
  static_assert(NT % 64 == 0, "one element per lane: (TX + 1)(TY + 1) must fill whole waves (the store phases stride by waves)");
};

// corner coordinates of the 2 x 2 nodes (j, i) above lattice node n00 of z-plane pl
__device__ __forceinline__ void march_load_plane(const LatArgs& T, int pl, int n00, double (&Q)[2][2][3]) {
  const double* q = T.xyz + (int64_t)(lat_plane<true>(T, pl) + n00) * 3;
  const int nx = T.nx;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) Q[j][i][c] = q[(j * nx + i) * 3 + c];
}

// Row offsets and Dirichlet flags of ONE plane of a column (the node box spans the planes l-1, l, l+1), in two steps as in
// the tile kernels (loads before the arithmetic, LDS commit after it) -- but branch-free: the plane bases are uniform
// scalars, every lane issues its NF byte loads back to back from clamped addresses and the flags are formed at commit.
template <int TX, int TY, int NT, int NDOF>
struct MarchMeta {
  using L = LatTile<TX, TY, 1>;
  static constexpr int NF = (L::NB + NT - 1) / NT, NRW = (L::NR + NT - 1) / NT;
  unsigned char raw[NF][NDOF];
  unsigned ok;       // bit j: flag j belongs to an existing node
  int r[NRW];
};

// the layer-independent half of the index arithmetic (a column keeps its x, y footprint while it marches): in-plane node
// offset / plane selector of every flag slot of this lane, in-row part of the closed-form row offset of its rows
template <int TX, int TY, int NT>
struct MarchMetaConst {
  using L = LatTile<TX, TY, 1>;
  static constexpr int NF = (L::NB + NT - 1) / NT, NRW = (L::NR + NT - 1) / NT;
  int off[NF];       // y nx + x of the flag's node, -1: outside the mesh / beyond the box
  int qz[NF];        // 0, 1, 2: plane l - 1, l, l + 1
  int c1[NRW];       // (3 y - (y > 0)) sx + cy (3 x - (x > 0)), -1: no such row
};

template <int TX, int TY, int NT>
__device__ __forceinline__ void march_meta_const(const LatArgs& T, int x0, int y0, int t, MarchMetaConst<TX, TY, NT>& K) {
  using L = LatTile<TX, TY, 1>;
  using MK = MarchMetaConst<TX, TY, NT>;
  const int nx = T.nx, ny = T.ny;
#pragma unroll
  for (int j = 0; j < MK::NF; ++j) {
    const int i = t + j * NT;
    const int qx = i % L::BX, qy = (i / L::BX) % L::BY;
    const int x = x0 - 1 + qx, y = y0 - 1 + qy;
    K.qz[j] = i / (L::BX * L::BY);
    K.off[j] = (i < L::NB && x >= 0 && x < nx && y >= 0 && y < ny) ? y * nx + x : -1;
  }
  const int sx = 3 * nx - 2;
#pragma unroll
  for (int j = 0; j < MK::NRW; ++j) {
    const int s = t + j * NT;
    const int x = x0 + s % TX, y = y0 + s / TX;
    const int cy = 3 - (y == 0) - (y == ny - 1);
    K.c1[j] = (s < L::NR && x < nx && y < ny) ? (3 * y - (y > 0)) * sx + cy * (3 * x - (x > 0)) : -1;
  }
}

template <int TX, int TY, int NT, int NDOF>
__device__ __forceinline__ void march_meta_load(const LatArgs& T, const MarchMetaConst<TX, TY, NT>& K, int l, MarchMeta<TX, TY, NT, NDOF>& M) {
  using MM = MarchMeta<TX, TY, NT, NDOF>;
  int base[3];
#pragma unroll
  for (int qz = 0; qz < 3; ++qz) {
    const int pl = T.p_own0 + l - 1 + qz;
    base[qz] = (pl >= 0 && pl < T.npl) ? lat_plane<true>(T, pl) : -1;   // uniform
  }
  M.ok = 0;
#pragma unroll
  for (int j = 0; j < MM::NF; ++j) {
    const int b = K.qz[j] == 0 ? base[0] : (K.qz[j] == 1 ? base[1] : base[2]);
    const bool ok = K.off[j] >= 0 && b >= 0;
    M.ok |= (ok ? 1u : 0u) << j;
    const int64_t node = ok ? (int64_t)b + K.off[j] : 0;
#pragma unroll
    for (int q = 0; q < NDOF; ++q) M.raw[j][q] = T.bcmask ? T.bcmask[node * NDOF + q] : (unsigned char)0;
  }
  // closed-form row offsets (lat_rowptr_std): (3 zo - (bot && zo > 0)) sy sx + cz c1 -- the marching kernels run on verified
  // index arithmetic only (std_lat)
  const int sx = 3 * T.nx - 2, sy = 3 * T.ny - 2;
  const bool bot = T.p_own0 == 0, top = T.p_own0 + T.n_own == T.npl;
  const int cz = 3 - (bot && l == 0) - (top && l == T.n_own - 1);
  const int az = (3 * l - (bot && l > 0)) * sy * sx;
#pragma unroll
  for (int j = 0; j < MM::NRW; ++j) M.r[j] = (K.c1[j] >= 0 && l < T.n_own) ? az + cz * K.c1[j] : -1;
}

template <int TX, int TY, int NT, int NDOF>
__device__ __forceinline__ int march_meta_commit(const LatArgs& T, int l, int t, const MarchMeta<TX, TY, NT, NDOF>& M, int* rlo, int* zrd,
                                                 unsigned char* nbc) {
  using L = LatTile<TX, TY, 1>;
  using MM = MarchMeta<TX, TY, NT, NDOF>;
  int any = 0;
#pragma unroll
  for (int j = 0; j < MM::NF; ++j)
    if (t + j * NT < L::NB) {
      unsigned char f = 0;
#pragma unroll
      for (int q = 0; q < NDOF; ++q) f |= (M.raw[j][q] && ((M.ok >> j) & 1u) ? 1 : 0) << q;
      nbc[t + j * NT] = f;
      any |= f;
    }
#pragma unroll
  for (int j = 0; j < MM::NRW; ++j)
    if (t + j * NT < L::NR) rlo[t + j * NT] = M.r[j];
  if (t == 0) zrd[0] = (l < T.n_own) ? lat_zcode<true>(T, T.p_own0 + l) : 0;
  return any;
}

// diagnostics (PYNAMA_MARCH_STAMPS): mean phase durations (shader cycles) over the layers of every workgroup, interior / edge columns
int march_print_stamps(pyn_ctx* c, LatArgs& T, int nblk, int ncol, int zlen) {
  std::vector<unsigned long long> h((size_t)nblk * 32 * 8);
  PYN_HIP(hipMemcpyAsync(h.data(), T.dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  double sum[2][8] = {{0}}, cnt[2] = {0, 0}, rt[2] = {0, 0};
  for (int b = 0; b < nblk; ++b) {
    const int col = b % ncol, bx = col % T.ntx, by = col / T.ntx;
    const int edge = (bx == 0 || by == 0 || bx == T.ntx - 1 || by == T.nty - 1) ? 1 : 0;
    for (int l = 2; l < std::min(zlen, 31); ++l) {
      const unsigned long long* q = &h[((size_t)b * 32 + l) * 8];
      const unsigned long long* qn = &h[((size_t)b * 32 + l + 1) * 8];
      if (!q[0] || !qn[0]) continue;
      for (int k = 0; k < 6; ++k) sum[edge][k] += (double)(q[k + 1] - q[k]);
      sum[edge][6] += (double)(qn[0] - q[0]);
      rt[edge] += (double)(qn[7] - q[7]);
      cnt[edge] += 1;
    }
  }
  for (int e = 0; e < 2; ++e)
    if (cnt[e] > 0)
      fprintf(stderr, "march stamps (%s columns, %g layer-steps): phases 0-1 %.0f | 1-2 %.0f | 2-3 %.0f | 3-4 %.0f | 4-5 %.0f | 5-6 %.0f | whole layer %.0f cycles = %.2f us "
              "(clock %.2f GHz)\n", e ? "edge" : "interior", cnt[e], sum[e][0] / cnt[e], sum[e][1] / cnt[e], sum[e][2] / cnt[e], sum[e][3] / cnt[e],
              sum[e][4] / cnt[e], sum[e][5] / cnt[e], sum[e][6] / cnt[e], rt[e] / cnt[e] / 100.0, sum[e][6] / (rt[e] * 10.0));
  T.dbg = nullptr;
  return PYN_OK;
}

// One workgroup = one (column of TX x TY rows) x (chunk of owned z-planes [zc0, zc1)); thread t integrates element
// (x0 - 1 + t % EX, y0 - 1 + t / EX) of every layer between planes zc0 - 1 and zc1.
#define MARCH_STAMP(k)                                                                                       \
  do {                                                                                                     \
    if (T.dbg && tid == 0 && l - zc0 + 1 < 32)                                                             \
      T.dbg[((size_t)blockIdx.x * 32 + (l - zc0 + 1)) * 8 + (k)] = __builtin_amdgcn_s_memtime();           \
  } while (0)

template <int TX, int TY, int WPS, int ROLLED>   // ROLLED: 0 pointwise unrolled, 1 pointwise rolled, 2 sum-factorised
__global__ void __launch_bounds__((TX + 1) * (TY + 1), WPS) assemble_q1_hex_march_kernel(LatArgs T, int zlen, double ws) {
  using MT = MarchTile<TX, TY>;
  using L1 = LatTile<TX, TY, 1>;
  constexpr int NT = MT::NT;
  extern __shared__ __align__(16) double lds[];
  int* rlo = reinterpret_cast<int*>(lds + 2 * MT::ACC);
  int* zrd = rlo + L1::NR;
  unsigned char* nbc = reinterpret_cast<unsigned char*>(zrd + 1);
  const int tid = threadIdx.x;
  const int ncol = T.ntx * T.nty;
  const int col = blockIdx.x % ncol, ch = blockIdx.x / ncol;
  const int x0 = (col % T.ntx) * TX, y0 = (col / T.ntx) * TY;
  const int zc0 = ch * zlen, zc1 = min(zc0 + zlen, T.n_own);
  const int lx = tid % MT::EX, ly = tid / MT::EX;
  const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
  const bool evalid = gx >= 0 && gx < T.nx - 1 && gy >= 0 && gy < T.ny - 1;
  const int n00 = evalid ? gy * T.nx + gx : 0;
  // LDS offsets of the rows of the four (j, i) corner columns, -1 when the row is not in this workgroup's column
  int ro[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rx = lx - 1 + i, ry = ly - 1 + j;
      ro[j][i] = (evalid && rx >= 0 && rx < TX && ry >= 0 && ry < TY) ? (ry * TX + rx) * 27 : -1;
    }
  MarchMetaConst<TX, TY, NT> mconst;
  march_meta_const<TX, TY, NT>(T, x0, y0, tid, mconst);
  for (int i = tid; i < 2 * MT::ACC; i += NT) lds[i] = 0.0;
  if (tid < 2) reinterpret_cast<int*>(nbc + ((L1::NB + 3) & ~3))[tid] = 0;
  double P[2][2][2][3] = {};   // [k][j][i][c]: bottom / top plane of the current layer
  double Pn[2][2][3] = {};     // top plane of the next layer (prefetch)
  {
    const int pb = T.p_own0 + zc0 - 1;
    if (evalid && pb >= 0) march_load_plane(T, pb, n00, P[0]);       // bottom plane of the first layer
    if (evalid && pb + 1 < T.npl) march_load_plane(T, pb + 1, n00, P[1]);
  }
  // retire the prologue loads HERE: otherwise the loop header inherits them as pending and its wait (vmcnt is one in-order
  // queue) also drains the previous layer's stores on every iteration
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  if (T.ablate >> 8) {   // diagnostics: stagger the workgroups' phases (units of 1024 cycles x (block % 4))
    const int n = (T.ablate >> 8) * (blockIdx.x & 3);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
  }
  __syncthreads();
  for (int l = zc0 - 1; l < zc1; ++l) {
    const int gl = T.p_own0 + l;                       // global index of the layer = of its bottom plane
    const bool lvalid = gl >= 0 && gl < T.npl - 1;     // (uniform) the layer exists
    const bool own_b = l >= zc0, own_t = l + 1 < zc1;  // (uniform) this workgroup owns the bottom / top plane rows
    MARCH_STAMP(0);
    march_load_plane(T, min(gl + 2, T.npl - 1), n00, Pn);   // unconditional (clamped): in flight during the arithmetic
    MarchMeta<TX, TY, NT, 1> meta;
    if (own_b) march_meta_load<TX, TY, NT, 1>(T, mconst, l, meta);
    const bool act = lvalid && evalid;
    double L[28];
    if (act && !(T.ablate & 1)) {
      if (ROLLED == 2) q1_laplace_sumfac(P, ws, L);
      else if (ROLLED == 1) q1_laplace_lean_rolled(P, ws, L);   // rolled loop over the Gauss points: one point's worth of registers
      else q1_laplace_lean(P, ws, L);
    }
    else {
#pragma unroll
      for (int i = 0; i < 28; ++i) L[i] = 0.0;
    }
    // retire the prefetch and the flag loads HERE, before this layer's stores are issued: a wave's loads and stores go
    // through one in-order counter, a wait placed after the stores (the compiler sinks the register rotation below into
    // the loop latch) would drain them on every layer.  The loads were issued a whole element integration ago.
    MARCH_STAMP(1);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    MARCH_STAMP(2);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          P[0][j][i][c] = P[1][j][i][c];
          P[1][j][i][c] = Pn[j][i][c];
        }
    __syncthreads();   // (A) the previous plane has left its buffer
    MARCH_STAMP(3);
    if (act && !(T.ablate & 2)) {
      double* bufb = lds + (l & 1) * MT::ACC;
      double* buft = lds + ((l + 1) & 1) * MT::ACC;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (!(k ? own_t : own_b)) continue;
        double* buf = k ? buft : bufb;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (ro[j][i] < 0) continue;
            const int a = Q1_NODE[i][j][k];
            double* row = buf + ro[j][i];
            double diag = 0.0;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
              for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                  const int b = Q1_NODE[i2][j2][k2];
                  if (b == a) continue;
                  const double v = q1_sym(L, a, b);
                  diag -= v;
                  atomicAdd(&row[(k2 - k + 1) * 9 + (j2 - j + 1) * 3 + (i2 - i + 1)], v);
                }
            atomicAdd(&row[13], diag);
          }
      }
    }
    int any = 0;
    if (own_b) any = march_meta_commit<TX, TY, NT, 1>(T, l, tid, meta, rlo, zrd, nbc);
    // (B) plane l is complete; "any imposed node in the box" through a parity-indexed LDS word (one barrier, not the three
    // of __syncthreads_or): slot l&1 is written before (B) and read after it, the other slot is cleared for the next layer
    int* anyflag = reinterpret_cast<int*>(nbc + ((L1::NB + 3) & ~3));
    if (__any(any) && (tid & 63) == 0) anyflag[l & 1] = 1;
    MARCH_STAMP(4);
    __syncthreads();
    MARCH_STAMP(5);
    const int anybc = anyflag[l & 1];
    if (tid == 0) anyflag[(l + 1) & 1] = 0;
    if (own_b && !(T.ablate & 8)) {
      double* buf = lds + (l & 1) * MT::ACC;
      lat_emit_dinv<TX, TY, 1>(T, x0, y0, l, buf, rlo, anybc ? nbc : nullptr, tid, NT);
      if (NT > 64 && T.dinv) __syncthreads();   // the stores below clear the accumulators other waves' lanes have just read
      if (lat_tile_plain<TX, TY, 1>(T, x0, y0, l, zrd, anybc) && T.ablate != 4)
        lat_store_plain<TX, TY, 1, true>(T, buf, rlo, tid, NT);
      else if (zrd[0] == ZCODE_STD && T.ablate != 4 && T.ablate != 16)   // boundary column, interior plane: masked x-line copies + the face rows
        lat_store_lines<TX, TY, true>(T, x0, y0, buf, rlo, zrd, nbc, anybc, tid, NT);
      else
        lat_store<TX, TY, 1, true>(T, x0, y0, buf, rlo, zrd, nbc, tid, NT);
    }
    MARCH_STAMP(6);
    if (T.dbg && tid == 0 && l - zc0 + 1 < 32) T.dbg[((size_t)blockIdx.x * 32 + (l - zc0 + 1)) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int TX, int TY, int WPS, int ROLLED = 0>
int launch_march(pyn_ctx* c, LatArgs& T, int wg_per_cu) {
  using MT = MarchTile<TX, TY>;
  T.ntx = (T.nx + TX - 1) / TX;
  T.nty = (T.ny + TY - 1) / TY;
  const int ncol = T.ntx * T.nty;
  // z-chunks: about 28 planes each, then as many chunks as fit into the same number of rounds of resident workgroups
  const int resident = 256 * wg_per_cu;
  const char* zl = getenv("PYNAMA_MARCH_ZLEN");
  int nzc = std::max(1, (T.n_own + 27) / 28);
  if (zl) nzc = std::max(1, (T.n_own + atoi(zl) - 1) / std::max(1, atoi(zl)));
  else {
    const int rounds = (ncol * nzc + resident - 1) / resident;
    nzc = std::min(T.n_own, std::max(nzc, rounds * resident / ncol));
  }
  const int zlen = (T.n_own + nzc - 1) / nzc;
  nzc = (T.n_own + zlen - 1) / zlen;
  static bool attr_done = false;
  if (!attr_done) {
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_march_kernel<TX, TY, WPS, ROLLED>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)MT::BYTES));
    attr_done = true;
  }
  const double ws = 1.0 / 512.0;   // unit weights of the 2x2x2 rule (checked: q1_gauss_standard)
  DevTmp stamps;
  const int nblk = ncol * nzc;
  if (getenv("PYNAMA_MARCH_STAMPS")) {
    PYN_HIP(stamps.alloc((size_t)nblk * 32 * 8 * sizeof(unsigned long long)));
    PYN_HIP(hipMemsetAsync(stamps.p, 0, (size_t)nblk * 32 * 8 * sizeof(unsigned long long), c->stream));
    T.dbg = stamps.as<unsigned long long>();
  }
  assemble_q1_hex_march_kernel<TX, TY, WPS, ROLLED><<<nblk, MT::NT, MT::BYTES, c->stream>>>(T, zlen, ws);
  PYN_HIP(hipGetLastError());
  if (T.dbg) PYN_TRY(march_print_stamps(c, T, nblk, ncol, zlen));
  return PYN_OK;
}

}  // namespace

// Are the uploaded full-rule tables those of the trilinear hexahedron at the 2x2x2 Gauss points, unit weights, corner
// order of SURVEY.md A.2 (any order of the points)?  The lean closed forms above replace the tables only then.
bool pyn_q1_gauss_tables_standard(const double* w, const double* H, const double* Hrs, const double* HrsCoo) {
  const int CO[3][8] = {{0, 0, 1, 1, 0, 1, 1, 0}, {0, 1, 1, 0, 0, 0, 1, 1}, {0, 0, 0, 0, 1, 1, 1, 1}};
  bool seen[8] = {false, false, false, false, false, false, false, false};
  for (int g = 0; g < 8; ++g) {
    if (fabs(w[g] - 1.0) > 1e-13) return false;
    // the point: xi_d = sum_a s_d(a) N_a
    double xi[3] = {0, 0, 0};
    for (int d = 0; d < 3; ++d)
      for (int a = 0; a < 8; ++a) xi[d] += (2 * CO[d][a] - 1) * H[g * 8 + a];
    int code = 0;
    for (int d = 0; d < 3; ++d) {
      if (fabs(fabs(xi[d]) - Q1_GP) > 1e-13) return false;
      code |= (xi[d] > 0 ? 1 : 0) << d;
    }
    if (seen[code]) return false;
    seen[code] = true;
    for (int a = 0; a < 8; ++a) {
      double p[3];
      for (int d = 0; d < 3; ++d) p[d] = 1.0 + (2 * CO[d][a] - 1) * xi[d];
      if (fabs(H[g * 8 + a] - p[0] * p[1] * p[2] / 8.0) > 1e-13) return false;
      for (int d = 0; d < 3; ++d) {
        const double h = (2 * CO[d][a] - 1) * p[(d + 1) % 3] * p[(d + 2) % 3] / 8.0;
        if (fabs(Hrs[g * 24 + d * 8 + a] - h) > 1e-13 || fabs(HrsCoo[g * 24 + d * 8 + a] - h) > 1e-13) return false;
      }
    }
  }
  return true;
}

// general-geometry scalar Laplacian on a lattice: z-marching kernel.  `tile`: 0 = default shape
int pyn_assemble_lattice_march(pyn_ctx* c, void* lat_args, int tile) {
  LatArgs& T = *static_cast<LatArgs*>(lat_args);
  switch (tile) {
    case 1: return launch_march<15, 11, 2>(c, T, 2);
    case 2: return launch_march<15, 7, 2>(c, T, 3);
    case 3: return launch_march<31, 7, 1>(c, T, 1);
    case 5: return launch_march<15, 15, 1>(c, T, 1);
    case 6: return launch_march<7, 7, 3, true>(c, T, 7);
    case 7: return launch_march<7, 7, 2>(c, T, 7);     // Gauss points unrolled: 256 VGPRs, 9 % slower than the rolled loop
    case 8: return launch_march<15, 7, 3, true>(c, T, 3);
    case 9: return launch_march<15, 7, 2, true>(c, T, 3);
    case 10: return launch_march<15, 15, 1, true>(c, T, 1);
    case 11: return launch_march<15, 11, 2, true>(c, T, 2);
    case 12: return launch_march<7, 7, 2, true>(c, T, 7);   // the default shape with the pointwise (rolled) element routine
    case 13: return launch_march<15, 7, 2, 2>(c, T, 3);
    case 14: return launch_march<15, 15, 1, 2>(c, T, 1);
    // one wave per workgroup, 7 workgroups per CU, rolled loop over the Gauss points (224 VGPRs): fastest measured
    default: return launch_march<7, 7, 2, 2>(c, T, 7);
  }
}
