// Shared pieces of the plan-free (lattice) kernels: descriptor, tile geometry, index arithmetic, tile meta data and the two
// store phases.  Included by pyn_assemble_lattice.hip (tile kernels, matrix-free products) and pyn_assemble_march.hip
// (z-marching general-geometry kernels).  See DESIGN.md 4/5.
#pragma once
#include "pyn_internal.h"
#include "pyn_q1_hex.h"

namespace {

// =================================================================================================
// Plan-free variant for meshes with STRUCTURED topology (box meshes: the reference's primary mesh,
// DMPlexDom.createBoxMesh, src/domain/dmplex.py:8-21; a rank's z-slab of one included).  Same scheme
// as the patch kernel -- a workgroup owns a TX x TY x TZ tile of rows, integrates every element touching
// it (one per lane), accumulates in LDS, writes each CSR row once -- but every index comes from integer
// arithmetic on the lattice descriptor instead of from HBM: no element list, no row-slot / scatter-map
// stream (84 B per patch-element in the plan), no dependent load chains in front of the stores.
// LDS accumulators use a fixed 27-point stencil layout acc[row][(dz+1)*9 + (dy+1)*3 + (dx+1)], so the
// LDS address of pair (a, b) is row_slot(a)*27 + a compile-time constant; the store phase maps CSR slot
// k of a row (columns sorted by node id: z-plane order from `zord`, then y, then x, clipped at the
// domain faces) back to the stencil position.
struct LatArgs {
  const double* xyz;
  const int32_t* rowptr;
  const uint8_t* bcmask;   // per node, may be null
  const int32_t* P;        // [npl] first node id of every z-plane
  const int32_t* zord;     // [npl]
  int nx, ny, npl, p_own0, n_own;
  int ntx, nty;            // tiles per direction
  int bz0, bzs;            // matrix-free products over a subset of the z-tiles: tile layer = bz0 + k * bzs
  int std_lat;             // P[j] == j*nx*ny, all planes owned, standard z order: plane bases, z codes AND row offsets
                           // come from arithmetic (no index loads at all)
  int lean;                // the uploaded tables are the standard 2x2x2 Gauss tables: closed-form element routine (q1_laplace_lean)
  int ablate;              // diagnostics (PYNAMA_LATTICE_ABLATE): 1 no element phase, 4 no plain-tile store path, 16 boundary columns of the march through the CSR-slot decode,
                           // 6 round-robin instead of XCD-contiguous tile order
  TileArgs q;              // quadrature tables (w, hrs, hcoo, aff) -- only those fields are used
  double* A;
  double* Arhs;
  double* dinv = nullptr;              // 1 / diagonal per owned row, written with the rows (may be null)
  int rhs_clean = 0;                   // Arhs already holds zeros wherever this Dirichlet set leaves zeros (DMat::rhs_clean): tiles without
                                       // an imposed node in their node box do not write their (all-zero) Arhs rows again
  const int32_t* rcrow = nullptr;      // Arhs is a COMPACT imposed-column matrix: first block of every owned node row in it, -1 = not stored
                                       // (honoured by the KLE lattice kernels; the scalar kernels are never handed a compact target)
  unsigned long long* dbg = nullptr;   // diagnostics (PYNAMA_MARCH_STAMPS): per-phase s_memtime stamps of the marching kernels
};

template <int TX, int TY, int TZ>
struct LatTile {
  static constexpr int NR = TX * TY * TZ, EX = TX + 1, EY = TY + 1, EZ = TZ + 1, NE = EX * EY * EZ;
  static constexpr int BX = TX + 2, BY = TY + 2, BZ = TZ + 2, NB = BX * BY * BZ;
  static constexpr int ACC = NR * 27;                                  // doubles
  static constexpr int META_INTS = NR + TZ + (NB + 3) / 4;             // rlo[NR], zrd[TZ], nbc[NB] bytes
  static constexpr size_t BYTES = ACC * sizeof(double) + META_INTS * sizeof(int);
};

// First node id of z-plane j.  Arithmetic form (std_lat): the owned planes carry the ids 0 .. n_owned-1 in z
// order, the ghost planes below them follow, then the ghost planes above (the numbering of a rank's z-slab; on one
// rank simply j*nx*ny).  pyn_lattice_detect verified that the uploaded numbering has this shape.
template <bool STD = false>   // STD: the caller has checked T.std_lat (no index loads compiled in)
__device__ __forceinline__ int lat_plane(const LatArgs& T, int j) {
  if (!STD && !T.std_lat) return T.P[j];
  const int pp = T.nx * T.ny, lo = T.p_own0, hi = T.p_own0 + T.n_own;
  if (j < lo) return (T.n_own + j) * pp;
  if (j >= hi) return (T.n_own + lo + (j - hi)) * pp;
  return (j - lo) * pp;
}

// CSR offset of the row of owned node (x, y, owned plane zo): rows in id order, len = cx cy cz with c = 3 minus
// the domain faces the node sits on (a slab interface is not a face: its ghost plane supplies the columns);
// sum_{x' < x} cx(x') = 3x - (x > 0), sum over a whole line = 3 nx - 2.  Verified against the graph's rowptr.
__device__ __forceinline__ int lat_rowptr_std(const LatArgs& T, int x, int y, int zo) {
  const int sx = 3 * T.nx - 2, sy = 3 * T.ny - 2;
  const bool bot = T.p_own0 == 0, top = T.p_own0 + T.n_own == T.npl;   // does the slab hold the domain's end planes?
  const int cy = 3 - (y == 0) - (y == T.ny - 1), cz = 3 - (bot && zo == 0) - (top && zo == T.n_own - 1);
  return (3 * zo - (bot && zo > 0)) * sy * sx + cz * ((3 * y - (y > 0)) * sx + cy * (3 * x - (x > 0)));
}

// z-order code of OWNED plane pl: the existing z-neighbours sorted by node id -- owned planes first (ascending),
// then the ghost plane below, then the ghost plane above
template <bool STD = false>
__device__ __forceinline__ int lat_zcode(const LatArgs& T, int pl) {
  if (!STD && !T.std_lat) return T.zord[pl];
  const int lo = T.p_own0, hi = T.p_own0 + T.n_own;
  const bool has_dn = pl > 0, has_up = pl < T.npl - 1;
  const bool dn_ghost = has_dn && pl - 1 < lo, up_ghost = has_up && pl + 1 >= hi;
  int code = 0, n = 0;
  if (has_dn && !dn_ghost) code |= 0 << (2 + 2 * n++);
  code |= 1 << (2 + 2 * n++);
  if (has_up && !up_ghost) code |= 2 << (2 + 2 * n++);
  if (dn_ghost) code |= 0 << (2 + 2 * n++);
  if (up_ghost) code |= 2 << (2 + 2 * n++);
  return code | n;
}

// Row offsets and Dirichlet flags of a tile, in two steps so that their HBM latency hides behind the element
// phase: lat_meta_load issues the loads into registers before it, lat_meta_commit writes them to LDS after it.
template <int TX, int TY, int TZ, int NT>
struct LatMeta {
  using L = LatTile<TX, TY, TZ>;
  static constexpr int NF = (L::NB + NT - 1) / NT, NRW = (L::NR + NT - 1) / NT;
  unsigned char f[NF];
  int r[NRW];
};

template <int TX, int TY, int TZ, int NT, int NDOF = 1>
__device__ __forceinline__ void lat_meta_load(const LatArgs& T, int x0, int y0, int z0, int t, LatMeta<TX, TY, TZ, NT>& M) {
  using L = LatTile<TX, TY, TZ>;
  const int nx = T.nx, ny = T.ny;
#pragma unroll
  for (int j = 0; j < LatMeta<TX, TY, TZ, NT>::NF; ++j) {
    const int i = t + j * NT;
    const int qx = i % L::BX, qy = (i / L::BX) % L::BY, qz = i / (L::BX * L::BY);
    const int x = x0 - 1 + qx, y = y0 - 1 + qy, pl = T.p_own0 + z0 - 1 + qz;
    const bool ok = T.bcmask && i < L::NB && x >= 0 && x < nx && y >= 0 && y < ny && pl >= 0 && pl < T.npl;
    unsigned char f = 0;   // bit q: DOF q of the node imposed
    if (ok) {
      const int64_t node = lat_plane(T, pl) + y * nx + x;
#pragma unroll
      for (int q = 0; q < NDOF; ++q) f |= (T.bcmask[node * NDOF + q] ? 1 : 0) << q;
    }
    M.f[j] = f;
  }
#pragma unroll
  for (int j = 0; j < LatMeta<TX, TY, TZ, NT>::NRW; ++j) {
    const int s = t + j * NT;
    const int rx = s % TX, ry = (s / TX) % TY, rz = s / (TX * TY);
    const int x = x0 + rx, y = y0 + ry, zo = z0 + rz;
    const bool ok = s < L::NR && x < nx && y < ny && zo < T.n_own;
    M.r[j] = !ok ? -1 : (T.std_lat ? lat_rowptr_std(T, x, y, zo) : T.rowptr[T.P[T.p_own0 + zo] + y * nx + x]);
  }
}

template <int TX, int TY, int TZ, int NT>
__device__ __forceinline__ int lat_meta_commit(const LatArgs& T, int z0, int t, const LatMeta<TX, TY, TZ, NT>& M, int* rlo, int* zrd,
                                               unsigned char* nbc) {
  using L = LatTile<TX, TY, TZ>;
  int any = 0;
#pragma unroll
  for (int j = 0; j < LatMeta<TX, TY, TZ, NT>::NF; ++j)
    if (t + j * NT < L::NB) {
      nbc[t + j * NT] = M.f[j];
      any |= M.f[j];
    }
#pragma unroll
  for (int j = 0; j < LatMeta<TX, TY, TZ, NT>::NRW; ++j)
    if (t + j * NT < L::NR) rlo[t + j * NT] = M.r[j];
  if (t < TZ) zrd[t] = (z0 + t < T.n_own) ? lat_zcode(T, T.p_own0 + z0 + t) : 0;
  return any;
}

// integrate every element touching the tile (one per lane) and add the rows the tile owns into acc
template <int TX, int TY, int TZ>
__device__ __forceinline__ void lat_integrate(const LatArgs& T, int x0, int y0, int z0, double* acc, int t0, int nt) {
  using LT = LatTile<TX, TY, TZ>;
  const int nx = T.nx, ny = T.ny;
  // corner offsets (dx, dy, dz) in the reference's closure order
  constexpr int CX[8] = {0, 0, 1, 1, 0, 1, 1, 0};
  constexpr int CY[8] = {0, 1, 1, 0, 0, 0, 1, 1};
  constexpr int CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  for (int t = t0; t < LT::NE; t += nt) {
    const int lx = t % LT::EX, ly = (t / LT::EX) % LT::EY, lz = t / (LT::EX * LT::EY);
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gl = T.p_own0 + z0 - 1 + lz;
    if (gx < 0 || gx >= nx - 1 || gy < 0 || gy >= ny - 1 || gl < 0 || gl >= T.npl - 1) continue;
    const int n00 = gy * nx + gx;
    const int pb = lat_plane(T, gl) + n00, pt = lat_plane(T, gl + 1) + n00;
    double X[8][3];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int node = (CZ[a] ? pt : pb) + CY[a] * nx + CX[a];
      const double* q = T.xyz + (int64_t)node * 3;
      X[a][0] = q[0];
      X[a][1] = q[1];
      X[a][2] = q[2];
    }
    double L[36];
    if (T.lean) {   // (uniform) closed form of the same 8-point rule: 1,900 instead of 2,540 FP64 instructions
      q1_laplace_lean36(X, L);
    } else if (T.q.aff && __all(element_is_affine(T.q, X) ? 1 : 0)) {
      affine_laplace(T.q, X, L);
    } else {
#pragma unroll
      for (int i = 0; i < 36; ++i) L[i] = 0.0;
#pragma nounroll
      for (int g = 0; g < 8; ++g) gauss_point(T.q, g, X, L);
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int rx = lx - 1 + CX[a], ry = ly - 1 + CY[a], rz = lz - 1 + CZ[a];
      // the row exists in x, y (the element does); in z it must be one of this tile's OWNED planes
      if (rx < 0 || rx >= TX || ry < 0 || ry >= TY || rz < 0 || rz >= TZ || z0 + rz >= T.n_own) continue;
      double* row = acc + ((rz * TY + ry) * TX + rx) * 27;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int kk = (CZ[c] - CZ[a] + 1) * 9 + (CY[c] - CY[a] + 1) * 3 + (CX[c] - CX[a] + 1);
        atomicAdd(&row[kk], L[tri(a, c)]);
      }
    }
  }
}

// Jacobi data on the way out: 1 / diagonal of every row of the tile, ONE lane per row and ONE division per lane (owned rows
// carry the ids (zo ny + y) nx + x; imposed rows are identity rows).  Called between the barrier that completes the
// accumulators and the store phase; `nbc` null: no imposed node in the node box.
template <int TX, int TY, int TZ>
__device__ __forceinline__ void lat_emit_dinv(const LatArgs& T, int x0, int y0, int z0, const double* acc, const int* rlo,
                                              const unsigned char* nbc, int t, int nt) {
  using LT = LatTile<TX, TY, TZ>;
  if (!T.dinv) return;
  for (int s = t; s < LT::NR; s += nt) {
    if (rlo[s] < 0) continue;
    const int rx = s % TX, ry = (s / TX) % TY, rz = s / (TX * TY);
    const bool fr = nbc && nbc[((rz + 1) * LT::BY + ry + 1) * LT::BX + rx + 1];
    T.dinv[((int64_t)(z0 + rz) * T.ny + (y0 + ry)) * T.nx + x0 + rx] = fr ? 1.0 : 1.0 / acc[s * 27 + 13];
  }
}

// write every row of the tile once (half a wave per row, UNROLL rows in flight): A gets the free columns,
// Arhs the imposed ones (negated), imposed rows become identity rows (mat_generator.py:113-118).
// ZERO: clear each accumulator after reading it (the z-marching kernels reuse the buffer for the next plane).
// The rows are map(0) .. map(n-1) (lat_store: all NR rows; the z-marching kernel's boundary columns: only the rows on a domain face).
template <int TX, int TY, int TZ, bool ZERO, typename RowMap>
__device__ __forceinline__ void lat_store_rows(const LatArgs& T, int x0, int y0, double* acc, const int* rlo, const int* zrd,
                                               const unsigned char* nbc, int t, int nt, int n, RowMap map) {
  using LT = LatTile<TX, TY, TZ>;
  const int nx = T.nx, ny = T.ny;
  double* __restrict__ outA = T.A;
  double* __restrict__ outR = T.Arhs;
  const int half = t >> 5, k = t & 31;
  const int NH = nt >> 5;
  constexpr int UNROLL = 4;
  // branch-free per row: the LDS reads of the UNROLL rows are independent of each other, so they overlap
  for (int s0 = half; s0 < n; s0 += NH * UNROLL) {
    int lo[UNROLL], ai[UNROLL], bi[UNROLL], bo[UNROLL];
    bool diag[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int s = map(min(s0 + u * NH, n - 1));
      const int rl = rlo[s];
      const int rx = s % TX, ry = (s / TX) % TY, rz = s / (TX * TY);
      const int x = x0 + rx, y = y0 + ry;
      const int zi = zrd[rz];
      const int cx = 3 - (x == 0) - (x == nx - 1), cy = 3 - (y == 0) - (y == ny - 1), cz = zi & 3;
      const int cc = cx * cy;
      const bool act = (s0 + u * NH < n) && rl >= 0 && k < cc * cz;
      const int kz = (k >= cc) + (k >= 2 * cc);
      const int r = k - kz * cc;
      const int ky = (r >= cx) + (r >= 2 * cx);
      const int kx = r - ky * cx;
      const int dz = act ? ((zi >> (2 + 2 * kz)) & 3) - 1 : 0;
      const int dy = act ? ky - (y != 0) : 0, dx = act ? kx - (x != 0) : 0;
      ai[u] = s * 27 + (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1);
      bi[u] = ((rz + 1) * LT::BY + ry + 1) * LT::BX + rx + 1;
      bo[u] = bi[u] + (dz * LT::BY + dy) * LT::BX + dx;
      diag[u] = dx == 0 && dy == 0 && dz == 0;
      lo[u] = act ? rl + k : -1;
    }
    double v[UNROLL];
    unsigned char fr[UNROLL], fc[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      v[u] = acc[ai[u]];
      fr[u] = nbc[bi[u]];
      fc[u] = nbc[bo[u]];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const double va = fr[u] ? (diag[u] ? 1.0 : 0.0) : (fc[u] ? 0.0 : v[u]);
      const double vr = fr[u] ? (diag[u] ? 1.0 : 0.0) : (fc[u] ? -v[u] : 0.0);
      if (lo[u] >= 0) {
        outA[lo[u]] = va;
        if (outR) outR[lo[u]] = vr;
        if (ZERO) acc[ai[u]] = 0.0;   // every accumulated slot is some row's CSR entry: this clears the whole buffer
      }
    }
  }
}

template <int TX, int TY, int TZ, bool ZERO = false>
__device__ __forceinline__ void lat_store(const LatArgs& T, int x0, int y0, double* acc, const int* rlo, const int* zrd,
                                          const unsigned char* nbc, int t, int nt) {
  lat_store_rows<TX, TY, TZ, ZERO>(T, x0, y0, acc, rlo, zrd, nbc, t, nt, LatTile<TX, TY, TZ>::NR, [](int j) { return j; });
}

// Store phase of a "plain" tile -- no row on a domain face, the three z-neighbour planes in ascending id order,
// no imposed node in the node box: CSR slot k of a row IS stencil position k, and the TX rows of an x-line are
// one contiguous run of TX*27 doubles both in LDS and in the CSR value array.  Straight coalesced copy.
constexpr int ZCODE_STD = 3 | (0 << 2) | (1 << 4) | (2 << 6);
template <int TX, int TY, int TZ, bool ZERO = false>
__device__ __forceinline__ void lat_store_plain(const LatArgs& T, double* acc, const int* rlo, int t, int nt) {
  constexpr int LINE = TX * 27, NL = TY * TZ, PER = (LINE + 63) / 64;
  double* __restrict__ outA = T.A;
  double* __restrict__ outR = T.Arhs;
  const int w = t >> 6, lane = t & 63, nw = nt >> 6;
  for (int l = w; l < NL; l += nw) {
    const int base = rlo[l * TX];
    double v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = lane + 64 * j;
      v[j] = (i < LINE) ? acc[l * LINE + i] : 0.0;
      if (ZERO && i < LINE) acc[l * LINE + i] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = lane + 64 * j;
      if (i < LINE) {
        outA[base + i] = v[j];
        if (outR && !T.rhs_clean) outR[base + i] = 0.0;
      }
    }
  }
}

// Store phase of one plane (TZ = 1) of a BOUNDARY column of the z-marching kernel whose z-neighbour planes are in standard order
// (zrd == ZCODE_STD: every plane but the domain's first and last).  Most of its rows still have the full 27-point stencil: on an
// x-line with interior y the rows x = 1 .. nx-2 are one contiguous run in LDS and in the CSR array with slot == stencil position,
// i.e. the x-line copy of lat_store_plain plus the Dirichlet routing (two flag bytes per entry) -- no CSR-slot decode.  Only the
// rows ON a domain face (x = 0, x = nx-1, y = 0, y = ny-1: one row per line or one line per plane) take the decode of lat_store_rows.
template <int TX, int TY, bool ZERO>
__device__ __forceinline__ void lat_store_lines(const LatArgs& T, int x0, int y0, double* acc, const int* rlo, const int* zrd,
                                                const unsigned char* nbc, int anybc, int t, int nt) {
  using LT = LatTile<TX, TY, 1>;
  const int nx = T.nx, ny = T.ny;
  double* __restrict__ outA = T.A;
  double* __restrict__ outR = T.Arhs;
  const int txv = min(TX, nx - x0), tyv = min(TY, ny - y0);                         // rows of the column inside the domain
  const int r0 = x0 == 0 ? 1 : 0, r1 = (x0 + txv == nx) ? txv - 1 : txv;            // [r0, r1): rows with interior x
  const int nrun = max(r1 - r0, 0) * 27;
  const int w = t >> 6, lane = t & 63, nw = nt >> 6;
  constexpr int PER = (TX * 27 + 63) / 64;
  for (int ry = w; ry < tyv; ry += nw) {
    const int y = y0 + ry;
    if (y == 0 || y == ny - 1 || nrun == 0) continue;
    const int base = rlo[ry * TX + r0];
    const int a0 = (ry * TX + r0) * 27, b0 = (LT::BY + ry + 1) * LT::BX + r0 + 1;
    // all LDS reads of the line first (independent of each other), then the routing, then the stores
    double v[PER];
    unsigned char fr[PER], fc[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = min(lane + 64 * j, nrun - 1);
      const int r = i / 27, kk = i - r * 27;
      const int dz = kk / 9, dy = (kk - dz * 9) / 3, dx = kk - dz * 9 - dy * 3;     // 0 .. 2 each
      v[j] = acc[a0 + i];
      fr[j] = anybc ? nbc[b0 + r] : 0;
      fc[j] = anybc ? nbc[b0 + r + ((dz - 1) * LT::BY + dy - 1) * LT::BX + dx - 1] : 0;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = lane + 64 * j;
      if (i < nrun) {
        const int kk = i % 27;
        const double idv = kk == 13 ? 1.0 : 0.0;
        if (ZERO) acc[a0 + i] = 0.0;
        outA[base + i] = fr[j] ? idv : (fc[j] ? 0.0 : v[j]);
        if (outR) outR[base + i] = fr[j] ? idv : (fc[j] ? -v[j] : 0.0);
      }
    }
  }
  // rows on a domain face: the x-end row of every interior-y line, then the whole y-face lines
  const bool xlo = x0 == 0, xhi = x0 + txv == nx, ylo = y0 == 0, yhi = y0 + tyv == ny;
  const int nyf = (ylo ? 1 : 0) + ((yhi && !(ylo && tyv == 1)) ? 1 : 0);               // y-face lines in this column
  const int nxe = (xlo ? 1 : 0) + ((xhi && !(xlo && txv == 1)) ? 1 : 0);               // x-end rows per line
  const int yin0 = ylo ? 1 : 0, nyin = tyv - nyf;                                      // interior-y lines [yin0, yin0 + nyin)
  const int nspecial = nyf * txv + max(nyin, 0) * nxe;
  if (nspecial > 0)
    lat_store_rows<TX, TY, 1, ZERO>(T, x0, y0, acc, rlo, zrd, nbc, t, nt, nspecial, [=](int j) {
      if (j < nyf * txv) {                       // y-face lines (at most two), row by row
        const int f = j >= txv ? 1 : 0, rx = j - f * txv;
        const int ry = (f == 0 && ylo) ? 0 : tyv - 1;
        return ry * TX + rx;
      }
      const int q = j - nyf * txv;               // x-end rows (one or two per line) of the interior-y lines
      const int ly = nxe == 2 ? q >> 1 : q, e = q - ly * nxe;
      const int rx = (e == 0 && xlo) ? 0 : txv - 1;
      return (yin0 + ly) * TX + rx;
    });
}

template <int TX, int TY, int TZ>
__device__ __forceinline__ bool lat_tile_plain(const LatArgs& T, int x0, int y0, int z0, const int* zrd, int anybc) {
  bool plain = !anybc && x0 >= 1 && x0 + TX <= T.nx - 1 && y0 >= 1 && y0 + TY <= T.ny - 1 && z0 + TZ <= T.n_own;
  for (int j = 0; j < TZ; ++j) plain = plain && zrd[j] == ZCODE_STD;
  return plain;
}

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Remap so that every XCD works on ONE
// contiguous range of tiles: neighbouring tiles share node coordinates / flags, which then hit the same L2.
__device__ __forceinline__ int xcd_contiguous_tile(int bid, int n) {
  const int q = n >> 3, r = n & 7, j = bid & 7, idx = bid >> 3;
  return j * q + (j < r ? j : r) + idx;
}

}  // namespace
