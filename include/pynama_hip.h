/* pynama_hip.h -- C ABI of libpynama_hip.so: the MI355X (gfx950) implementation of the
 * Pynama finite/spectral-element hot path (element quadrature + global scatter + Krylov solve).
 *
 * Plain C: opaque context, plain pointers and sizes, int32 indices, float64 values.  No
 * PyTorch / C++ types cross this boundary.  Every function returns 0 on success and a
 * negative PYN_E* code on failure; the message is available from pyn_last_error().
 * Host arrays are caller-owned, C-contiguous, borrowed for the duration of the call.
 * Device memory is owned by the context.  A context is bound to ONE GPU and ONE process
 * (one process per GPU; ranks are tied together with pyn_comm_init over RCCL).  A context is
 * not re-entrant.
 *
 * Each entry point names the reference interface (file:line under /root/reference/) whose
 * work it takes over.  The reference has no FFI: the boundary sits behind its Python classes
 * (Spectral, DMPlexDom, Mat, KspSolver, FreeSlip); pynama_amd/ mirrors those classes on top of
 * this header with ctypes.
 */
#ifndef PYNAMA_HIP_H
#define PYNAMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pyn_ctx pyn_ctx;

enum {
  PYN_OK = 0,
  PYN_EINVAL = -1,   /* bad argument / wrong call order */
  PYN_EHIP = -2,     /* HIP runtime error */
  PYN_ENCCL = -3,    /* RCCL error */
  PYN_ENOTCONV = -4, /* reserved */
  PYN_ENOGPU = -5    /* no gfx950 device visible */
};

/* quadrature table slots -- src/elements/spectral.py:45-61 (H/Hrs/gps, HRed/..., HOp/...) */
enum { PYN_Q_FULL = 0, PYN_Q_RED = 1, PYN_Q_NODAL = 2 };

/* scalar forms for pyn_assemble_scalar / pyn_elem_local */
enum {
  PYN_FORM_LAPLACE = 0,   /* L_e = sum_full w detJ G^T G          spectral.py:125,131 (one component) */
  PYN_FORM_MASS_NODAL = 1,/* M_e = sum_nodal w detJ H H^T         spectral.py:215 (elWeigMat)         */
  PYN_FORM_MASS_FULL = 2, /* same on the full rule (consistent mass)                                   */
  PYN_FORM_KLE = 3,       /* K_e, Rw_e, Rd_e                       spectral.py:89-157                  */
  PYN_FORM_OPERATOR = 4   /* first-order operator blocks (internal to pyn_assemble_operator)            */
};

/* Krylov method / preconditioner / norm (PETSc option names in comments) */
enum { PYN_KSP_CG = 0, PYN_KSP_GMRES = 1 };                 /* -ksp_type cg | gmres            */
enum { PYN_PC_NONE = 0, PYN_PC_JACOBI = 1 };                /* -pc_type none | jacobi          */
enum { PYN_NORM_PRECONDITIONED = 0, PYN_NORM_UNPRECONDITIONED = 1, PYN_NORM_NATURAL = 2 };
/* converged reasons, numbered as PETSc's KSPConvergedReason */
enum { PYN_CONVERGED_RTOL = 2, PYN_CONVERGED_ATOL = 3, PYN_CONVERGED_ITS = 4,
       PYN_DIVERGED_ITS = -3, PYN_DIVERGED_DTOL = -4, PYN_DIVERGED_BREAKDOWN = -5,
       PYN_DIVERGED_NANORINF = -9 };

const char* pyn_last_error(void);
int pyn_version(void);
const char* pyn_source_hash(void);   /* first 16 hex digits of the sha256 of the kernel sources this library was built from */
int pyn_device_count(int* count);

/* ---- context -------------------------------------------------------------------------- */
int pyn_ctx_create(int device, pyn_ctx** out);
int pyn_ctx_destroy(pyn_ctx* ctx);
int pyn_sync(pyn_ctx* ctx);                       /* hipStreamSynchronize on the context stream */

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) ------------------------------------
 * Replaces the MPI communicator the reference hands to PETSc (src/cases/base_problem.py:22,
 * src/matrices/mat_generator.py:95-99) and the implicit collectives inside MatAssembly /
 * KSPSolve. */
int pyn_comm_unique_id(void* out, int nbytes);    /* rank 0: nbytes >= 128 */
/* unique_id == NULL with nranks > 1 declares the ranks WITHOUT a transport ("detached"): the rank's
 * slab can be assembled and multiplied in isolation, ghost entries being supplied by the caller
 * through pyn_vec_set_local_host; collectives and pyn_solve are refused.
 * nranks == 1: unique_id == NULL means serial (no communicator, no RCCL calls); a unique id creates a
 * real one-rank RCCL communicator, so the collective code paths (and a halo plan whose neighbour is
 * the rank itself) run exactly as they do at nranks > 1. */
int pyn_comm_init(pyn_ctx* ctx, int rank, int nranks, const void* unique_id, int nbytes);
/* TEST transport: the same collectives staged through a POSIX shared-memory file (device -> host, process barrier,
 * host -> device), so that world_size > 1 can be exercised end to end by processes sharing ONE GPU, which RCCL refuses
 * (duplicate device).  `path` names a zero-filled file of at least 4096 + nranks*512 + nranks*nranks*16 + nranks*cap_bytes
 * bytes created by the launcher; cap_bytes bounds one rank's packed halo.  Never used by bench.py or the cases. */
int pyn_comm_init_shm(pyn_ctx* ctx, int rank, int nranks, const char* path, int64_t cap_bytes);
int pyn_comm_barrier(pyn_ctx* ctx);               /* device + host barrier over all ranks */
int pyn_comm_allreduce_f64(pyn_ctx* ctx, double* inout, int n, int op /*0 sum, 1 max*/);
/* Start-up self-test of the communicators (all-reduces and halo exchanges have one each): RCCL's own rank count of both, all-reduce
 * of 1 and of the rank, one halo exchange of a rank-stamped vector on the main stream, one on the communication stream (the
 * overlapped form of the CG), and one on the communication stream WHILE an all-reduce is queued on the main stream, each checked on
 * the receiver.  info[6]: ranks counted by RCCL (0: the shared-memory test transport), sum(1), sum(rank), ghosts checked (main /
 * communication stream), sum(rank + 1) of the all-reduce that ran beside an exchange, [6] 1 = the halo exchanges have a communicator of
 * their own, 2 = they share the all-reduces' (RCCL refused the split; reported on stderr).  The
 * reference's analogue is implicit: PETSc checks its communicator at KSPSetUp / MatAssemblyEnd (src/solver/ksp_solver.py:19,
 * src/matrices/mat_generator.py:14-17).  Call after pyn_halo_set. */
int pyn_comm_selftest(pyn_ctx* ctx, double* info, int ninfo);

/* Row partition + halo plan of this rank.  Local node numbering: owned nodes
 * [0, n_owned) in global order, then ghosts grouped by owning neighbour (recv order).
 *   send_idx[send_ptr[k] .. send_ptr[k+1]) : owned local node ids sent to neigh[k]
 *   ghosts from neigh[k] occupy local ids n_owned + recv_ptr[k] .. n_owned + recv_ptr[k+1]
 * Must be called BEFORE pyn_mesh_set when nranks > 1 (default: everything owned). */
int pyn_halo_set(pyn_ctx* ctx, int64_t n_owned, int64_t n_ghost, int n_neigh, const int32_t* neigh,
                 const int64_t* send_ptr, const int32_t* send_idx, const int64_t* recv_ptr);

/* ---- mesh, element tables, boundary condition ------------------------------------------- */
/* Connectivity + coordinates of the LOCAL mesh (owned + ghost nodes).  Takes over
 * DMPlexDom.getCellCornersCoords (src/domain/dmplex.py:97-104) and getGlobalNodesFromCell
 * (dmplex.py:197-200 -> src/domain/indices.py:66-88) for every cell at once:
 * conn[e*nn + a] = local node id of element-local node a (reference order; the
 * first 2^dim entries are the corners in DMPlex closure order), xyz[n*dim + d].
 * nn == dim + 1 selects linear simplices (triangles / tetrahedra; no counterpart in the reference, which
 * is tensor-product only -- BASELINE.json configs[4]): the geometry nodes are the element's own nodes,
 * HrsCoo has dim+1 columns, and every entry point below works unchanged. */
int pyn_mesh_set(pyn_ctx* ctx, int dim, int nn, int64_t n_elem, int64_t n_node,
                 const int32_t* conn, const double* xyz);
/* The reference's box mesh built where it is used: replaces PETSc.DMPlex().createBoxMesh(faces, lower, upper)
 * (src/domain/dmplex.py:16-21) + the per-cell closure / coordinate interpolation of setFemIndexing / computeFullCoordinates
 * (dmplex.py:42-95) for this rank's block of a structured mesh of ngl^dim-node cells, without a host copy of either array:
 *   nel_local[dim]  elements per axis of the block (only the slowest axis -- y in 2-D, z in 3-D -- may be cut),
 *   layer0          first global element layer of the block along the slowest axis,
 *   lattice[dim]    nodes per axis of the WHOLE mesh ((ngl-1) * elements + 1),
 *   loc[nn*dim]     lattice offset in 0..ngl-1 of local node a along axis d (the reference's local order),
 *   planes[n_planes] global slowest-axis index of local plane k: local node id = k * (nodes per plane) + in-plane id, owned planes
 *                   first (pyn_halo_set's numbering); n_planes = (ngl-1) * nel_local[slow] + 1,
 *   axes            coordinate of lattice line i of axis d, the axes one after the other (lattice[0] + lattice[1] (+ lattice[2]) doubles).
 * Everything pyn_mesh_set does afterwards (topology detection, verified against the generated connectivity) is the same. */
int pyn_mesh_box(pyn_ctx* ctx, int dim, int ngl, const int64_t* nel_local, int64_t layer0, const int64_t* lattice,
                 const int32_t* loc, int64_t n_planes, const int64_t* planes, const double* axes);
/* Host copies of the local mesh as the device holds it (either pointer may be NULL): conn[n_elem*nn], xyz[n_node*dim].
 * DMPlexDom.getCellCornersCoords / getNodesCoordinates (dmplex.py:97-104, 230-244) for a caller that wants them all. */
int pyn_mesh_get(pyn_ctx* ctx, int32_t* conn, double* xyz);
/* Topology recognised by pyn_mesh_set: kind 0 = general connectivity, 1 = structured lattice of Q1
 * hexahedra (the reference's box mesh, src/domain/dmplex.py:8-21, or a rank's z-slab of one): nx, ny =
 * nodes per x / y line, nz = node planes of the local mesh.  Lattices are assembled by a plan-free kernel.
 * kind 2 = structured mesh of SECOND-order cells (ngl = 3: 9-node quadrilaterals / 27-node hexahedra in the
 * reference's local order, src/elements/spectral.py:346-431 -- the order of every yaml file under src/cases),
 * kind 3 = structured mesh of Q1 quadrilaterals; for both nx, ny, nz count the NODES per x-line, x-lines per
 * plane and planes (2-D: nz = 1) and the atomics-free row-run kernels assemble them (pyn_assemble_ho3.hip). */
int pyn_mesh_topology(pyn_ctx* ctx, int* kind, int* nx, int* ny, int* nz);
/* One quadrature's tables -- Spectral.computeMats2D/3D output (spectral.py:220-344):
 * w[ngp], H[ngp*nn], Hrs[ngp*dim*nn], HrsCoo[ngp*dim*2^dim] (geometry basis, spectral.py:54-61). */
int pyn_elem_tables_set(pyn_ctx* ctx, int which, int ngp, const double* w, const double* H,
                        const double* Hrs, const double* HrsCoo);
/* Dirichlet mask per LOCAL velocity DOF (node*ndof + d): 1 = value imposed.  Takes over the
 * set algebra of FreeSlip.buildKLEMats (src/cases/base_problem.py:512-528).  ndof = dim for the
 * KLE forms, 1 for scalar forms.  mask == NULL clears it. */
int pyn_bc_set(pyn_ctx* ctx, int ndof, const uint8_t* mask);

/* ---- symbolic phase ----------------------------------------------------------------------
 * Node adjacency graph -> CSR row pointers / column indices on the device (rows = owned
 * nodes, columns = local node ids, sorted).  Takes over DMPlexDom.getMatIndices
 * (dmplex.py:305-333) and the nnz preallocation of Mat.createEmptyKLEMats
 * (src/matrices/mat_generator.py:32-99).  All matrices share this one graph; a matrix with
 * block shape (br, bc) stores, for node-row i and component p, the scalar row
 * [(i,p) ; (col_k, q)] contiguously:  val[(rowptr[i]*br + p*len_i + k)*bc + q]. */
int pyn_csr_symbolic(pyn_ctx* ctx);
int pyn_csr_info(pyn_ctx* ctx, int64_t* n_rows, int64_t* nnz_blocks);
int pyn_csr_get(pyn_ctx* ctx, int32_t* rowptr, int32_t* colidx);

/* Optional patch plan for the atomics-free tiled assembly (Q1 hexahedra): a partition of the owned
 * rows into patches of <= 352 rows, patch p owning patch_rows[patch_ptr[p] .. patch_ptr[p+1]).  One
 * workgroup per patch integrates every element touching its rows, accumulates in LDS and writes each
 * CSR row once (no HBM atomics, no zero fill).  Built on the device from the connectivity; call after
 * pyn_csr_symbolic.  n_patch == 0 removes the plan.  Meshes without a plan use the generic kernel. */
int pyn_patch_plan_set(pyn_ctx* ctx, int n_patch, const int32_t* patch_ptr, const int32_t* patch_rows);
/* kind 0: the plan of the scalar forms (<= 352 rows per patch); kind 1: the plan of the tiled KLE
 * assembly (3x3 blocks: <= 36 rows per patch, e.g. 4x3x3 node tiles).  Both may coexist. */
int pyn_patch_plan_set_kind(pyn_ctx* ctx, int kind, int n_patch, const int32_t* patch_ptr, const int32_t* patch_rows);
/* The plan in use (the caller's, or the automatic one an assembly built): info[4] = patches, longest patch (rows), longest
 * row of the graph (entries), (patch, element) pairs -- pairs / elements is the factor by which elements on patch borders are
 * integrated more than once (diagnostics; no counterpart in the reference). */
int pyn_patch_plan_info(pyn_ctx* ctx, int kind, int64_t* info);

/* ---- matrices and vectors (device resident) ---------------------------------------------
 * Handles are small non-negative ints.  A vector with block size b has (n_owned+n_ghost)*b
 * entries; only the owned part is meaningful to the caller. */
int pyn_mat_create(pyn_ctx* ctx, int br, int bc, int* mat_id);      /* mat_generator.py:95-99 */
/* Krhs / Krhsfs / Arhs ("imposed-column" matrices: -K_e[free, bc] and the unit diagonal of the imposed DOFs, zero elsewhere) in COMPACT
 * form: only the node rows with an imposed node in their neighbourhood are stored -- the preallocation the reference makes for Krhs
 * (src/matrices/mat_generator.py:42-58, 91: `drhs_nnz` counts the Dirichlet columns of a row).  Laid out for the Dirichlet set current
 * at creation (pyn_bc_set first); an assembly under another set lays it out again.  Valid as the Krhs / Krhsfs / Arhs argument of the
 * assemblies, in pyn_spmv (rows that are not stored are zero rows), pyn_mat_add_values (entries in stored rows), pyn_mat_zero,
 * pyn_mat_get_values (the graph's full layout is returned) and pyn_mat_destroy; not a system matrix for pyn_solve.  A matrix made by
 * pyn_mat_create serves as Krhs as well (full pattern: 11 x the memory at 128^3, every product streams it whole). */
int pyn_mat_create_rhs(pyn_ctx* ctx, int br, int bc, int* mat_id);
/* what the matrix really stores: graph blocks (x br x bc doubles) and node rows -- Mat.getInfo() / printMatsInfo (mat_generator.py:120-130) */
int pyn_mat_stored_blocks(pyn_ctx* ctx, int mat_id, int64_t* blocks, int64_t* node_rows);
int pyn_mat_destroy(pyn_ctx* ctx, int mat_id);                      /* Mat.destroy(): values, solver image and Jacobi data are released, the handle dies */
int pyn_mat_zero(pyn_ctx* ctx, int mat_id);
/* Host insertion path, Mat.setValues(rows, cols, vals, addv) (src/cases/base_problem.py:531-547, src/matrices/mat_generator.py:
 * 113-118, 157-170): scalar DOF indices (node * block + component, LOCAL numbering), vals row-major [nrows][ncols];
 * insert != 0: INSERT_VALUES.  Entries outside the node graph are an error; rows of other ranks are dropped. */
int pyn_mat_add_values(pyn_ctx* ctx, int mat_id, int nrows, const int32_t* rows, int ncols, const int32_t* cols, const double* vals,
                       int insert);
int pyn_mat_get_values(pyn_ctx* ctx, int mat_id, double* val);      /* layout above */
int pyn_mat_get_diagonal(pyn_ctx* ctx, int mat_id, int vec_id);
int pyn_mat_axpy(pyn_ctx* ctx, int y_mat, double a, int x_mat);     /* Y += a X (base_problem.py:318) */
int pyn_mat_row_scale(pyn_ctx* ctx, int mat_id, int vec_id);        /* diagonalScale(L=) mat_generator.py:176; a vector of block size 1
                                                                      * holds one factor per NODE for all of its rows (the lumped weights) */
int pyn_vec_create(pyn_ctx* ctx, int bs, int* vec_id);
int pyn_vec_destroy(pyn_ctx* ctx, int vec_id);
int pyn_vec_set_host(pyn_ctx* ctx, int vec_id, const double* src);  /* owned part, n_owned*bs */
int pyn_vec_set_local_host(pyn_ctx* ctx, int vec_id, const double* src); /* owned + ghost, (n_owned+n_ghost)*bs */
int pyn_vec_get_host(pyn_ctx* ctx, int vec_id, double* dst);
int pyn_vec_fill(pyn_ctx* ctx, int vec_id, double value);
int pyn_vec_scatter_host(pyn_ctx* ctx, int vec_id, int64_t n, const int32_t* idx, const double* vals,
                         int add);                                   /* Vec.setValues */
/* w = a*x + b*y (x,y,w may alias) ; w = x.*y ; w = 1./x ; reductions over owned entries,
 * all-reduced across ranks */
int pyn_vec_axpby(pyn_ctx* ctx, int w, double a, int x, double b, int y);
int pyn_vec_pointwise_mult(pyn_ctx* ctx, int w, int x, int y);
int pyn_vec_reciprocal(pyn_ctx* ctx, int x);
int pyn_vec_vtensv(pyn_ctx* ctx, int v, int out);   /* v (x) v, BaseProblem.computeVtensV (base_problem.py:234-252) */
int pyn_vec_dot(pyn_ctx* ctx, int x, int y, double* out);
int pyn_vec_norm(pyn_ctx* ctx, int x, int type /*1, 2, 3=inf (PETSc NormType)*/, double* out);

/* ---- numeric phase (HOT LOOP 1) -----------------------------------------------------------
 * One device pass over all local elements: quadrature (spectral.py:89-157) + scatter-add with
 * Dirichlet elimination (base_problem.py:499-552) + unit diagonal on imposed DOFs
 * (mat_generator.py:113-118).  Matrices are zeroed first.  Any id may be -1 (skipped).
 *   K   [dim,dim]  += K_e[free,free]         Krhs[dim,dim] += -K_e[free,bc]
 *   Rw  [dim,dim_w]+= Rw_e[free,:]           Rd  [dim,1]   += Rd_e[free,:]
 * variant: 0 = workgroup-per-element + FP64 atomics (any mesh, any ngl), 1 = auto: the tiled atomics-free
 * kernels for Q1 hexahedra (with the caller's patch plan, else patches of consecutive rows), else 0. */
int pyn_assemble_kle(pyn_ctx* ctx, double alpha_d, double alpha_w, int K, int Krhs, int Rw, int Rd,
                     int variant);
/* No-slip / free-slip split of the same pass (NoSlipFreeSlip.buildKLEMats, src/cases/base_problem.py:
 * 329-454; MatNS, src/matrices/mat_ns.py:17-121).  pyn_bc_set(dim, cls) holds a CLASS per velocity DOF:
 * 0 free, 1 tangential DOF at a no-slip wall (free in the free-slip solve, imposed in the final one),
 * 2 imposed in both.  mat_ids[8] = K, Krhs, Rw, Rd, Kfs, Krhsfs, Rwfs, Rdfs (-1 = skipped):
 *   K[0,0] += v        Krhs[0,{1,2}] += -v       Rw[0,:], Rd[0,:]          unit diagonal on classes 1, 2
 *   Kfs[1,{0,1}] , Kfs[0,1] += v  (diag -1 on class 1)      Krhsfs[{0,1},2] += -v  (diag 1 on class 2)
 *   Rwfs[1,:], Rdfs[1,:] */
int pyn_assemble_kle_noslip(pyn_ctx* ctx, double alpha_d, double alpha_w, const int* mat_ids);
/* Scalar forms with the same elimination rule: A[1,1] += A_e[free,free], Arhs += -A_e[free,bc]. */
int pyn_assemble_scalar(pyn_ctx* ctx, int form, int A, int Arhs, int variant);
/* Single-element entry used for fixture parity: runs the SAME device element routine on one
 * element given by its corner coordinates and returns dense row-major matrices
 * (spectral.py:89-157 signature: coords -> K_e, Rw_e, Rd_e; any of the outputs may be NULL).
 * PYN_FORM_KLE: out0 = K_e[dim nn, dim nn], out1 = Rw_e[dim nn, dim_w nn], out2 = Rd_e[dim nn, nn];
 * scalar forms: out0 = A_e[nn, nn]. */
int pyn_elem_local(pyn_ctx* ctx, int form, double alpha_d, double alpha_w, const double* corners,
                   double* out0, double* out1, double* out2);

/* First-order operator blocks of Spectral.getElemKLEOperators (src/elements/spectral.py:159-218:
 * SrT, DivSrT, Curl) and their global scatter (Operators.setValues, src/matrices/mat_generator.py:
 * 157-170).  With G_g = J^-1 Hrs the physical gradients at point g of quadrature `rule`, c_g = w detJ:
 *   M[(a,p),(b,q)] = sum_g c_g H_g[a] * sum_t [row_t == p and col_t == q] coef_t * G_g[der_t][b]
 * terms[t] = (row component, column component, derivative axis); the block shape is the matrix's.
 * No Dirichlet elimination (the reference applies none to the operators). */
int pyn_assemble_operator(pyn_ctx* ctx, int rule, int nterms, const int32_t* terms, const double* coef, int mat_id);
/* single element, dense row-major out[br*nn][bc*nn] (fixture parity) */
int pyn_elem_operator_local(pyn_ctx* ctx, int rule, int br, int bc, int nterms, const int32_t* terms,
                            const double* coef, const double* corners, double* out);

/* ---- SpMV and Krylov solve (HOT LOOP 2) ---------------------------------------------------
 * y = A x with halo exchange of x over RCCL when nranks > 1 (PETSc MatMult,
 * base_problem.py:481 "Rw*vort + Krhs*vel"). */
int pyn_spmv(pyn_ctx* ctx, int mat_id, int x_vec, int y_vec);
/* Matrix-free operators: y = A x WITHOUT an assembled matrix (PETSc analogue: a MATSHELL).  Element matrices are recomputed
 * on the fly (Spectral.getElemKLEMatrices, spectral.py:120-153) and applied per element; needs a Q1 hexahedral mesh with
 * structured topology (pyn_mesh_topology == lattice), errors otherwise.
 *   PYN_MATFREE_LAPLACE  the scalar Laplacian pyn_assemble_scalar(PYN_FORM_LAPLACE) builds (1 DOF per node)
 *   PYN_MATFREE_KLE      the K of pyn_assemble_kle (3 DOFs per node); alpha_d / alpha_w are that call's penalty weights
 *                        (1e3 / 1e2 in the reference, spectral.py:152-153)
 * pyn_matfree_set defines the operator from the mesh, the element tables and a SNAPSHOT of the current Dirichlet mask
 * (imposed rows identity, imposed columns eliminated, base_problem.py:531-549): call it next to the assembly it mirrors;
 * later pyn_bc_set calls do not change it. */
enum { PYN_MATFREE_OFF = 0, PYN_MATFREE_LAPLACE = 1, PYN_MATFREE_KLE = 2 };
int pyn_matfree_set(pyn_ctx* ctx, int op, double alpha_d, double alpha_w);
int pyn_matfree_apply(pyn_ctx* ctx, int op, int x_vec, int y_vec);
typedef struct pyn_solve_opts {
  int method;        /* PYN_KSP_*  */
  int pc;            /* PYN_PC_*   */
  int norm_type;     /* PYN_NORM_* */
  int maxit;         /* PETSc default 10000 */
  int restart;       /* GMRES(m), PETSc default 30 */
  int fixed_iters;   /* >0: run exactly this many iterations, no convergence exit (benchmarking) */
  int profile;       /* !=0: bracket every SpMV launch with HIP events (first 256 iterations) */
  int cg_variant;    /* 0 auto, 1 standard PCG, 2 single-reduction PCG (Chronopoulos-Gear; default for nranks>1) */
  int gmres_orthog;  /* 0 classical Gram-Schmidt + one refinement pass (-ksp_gmres_cgs_refinement_type refine_always),
                        1 classical without refinement (refine_never, PETSc's own default), 2 modified Gram-Schmidt
                        (-ksp_gmres_modifiedgramschmidt) */
  int matfree;       /* PYN_MATFREE_*: CG / GMRES multiply with the matrix-free operator instead of the assembled matrix, which
                        then only supplies the Jacobi diagonal and the exit check (KSPSetOperators(Amat = shell, Pmat =
                        assembled)); pyn_solve first verifies on b that both operators agree */
  double rtol, atol, dtol;   /* PETSc defaults 1e-5, 1e-50, 1e5 */
} pyn_solve_opts;
typedef struct pyn_solve_info {
  int iters;
  int reason;
  double rnorm;       /* last residual norm in the solver's norm type */
  double rnorm0;
  double true_resid;  /* ||b - A x||_2 / ||b||_2 recomputed at exit; -1 (not computed) when opts.fixed_iters > 0 */
  double solve_ms;    /* device time of the iteration loop (HIP events) */
  double spmv_ms;     /* mean device time of one SpMV launch (profile != 0), else 0 */
  int spmv_launches;  /* launches averaged in spmv_ms */
  double reduce_ms;   /* single-reduction CG with profile != 0: mean device time from the end of the product to the scalars being
                         ready (partial sums + all-reduce + scalar step), else 0 */
  double halo_ms;     /* same runs, overlapped exchange: mean device time of pack + grouped send/recv on the communication
                         stream (hidden behind the interior rows when shorter than their product), else 0 */
} pyn_solve_info;
/* Solve A x = b (x0 = 0).  Takes over KspSolver.createSolver + KSP.__call__
 * (src/solver/ksp_solver.py:9-19, call site base_problem.py:481). */
int pyn_solve(pyn_ctx* ctx, int mat_id, int b_vec, int x_vec, const pyn_solve_opts* opts,
              pyn_solve_info* info);
/* Direct solve of a SMALL system: blocked dense LU with partial pivoting, the factors cached in the matrix until its values change.
 * Stands for the reference's hard-wired `-ksp_type preonly -pc_type lu` (src/solver/ksp_solver.py:13-16, makefile:7: PETSc
 * factors at KSPSetUp, every later call is two triangular solves) at the sizes the reference's own tests use it
 * (src/tests/test_solver.py).  One rank, rows <= pyn_direct_max_rows(); info->iters = 1 and info->reason =
 * PYN_CONVERGED_ITS as PETSc reports for preonly; info->true_resid as in pyn_solve.  A zero pivot is PYN_EINVAL. */
int pyn_direct_max_rows(void);
int pyn_solve_direct(pyn_ctx* ctx, int mat_id, int b_vec, int x_vec, pyn_solve_info* info);

/* ---- timers -------------------------------------------------------------------------------
 * Device time (HIP events on the context stream) of the last call of each phase, in ms.
 * Replaces the tic/toc log of src/run_case.py:156-162. */
enum { PYN_T_SYMBOLIC = 0, PYN_T_ASSEMBLE = 1, PYN_T_SPMV = 2, PYN_T_SOLVE = 3, PYN_T_COUNT = 8 };
int pyn_timers_get(pyn_ctx* ctx, double* ms, int n);

#ifdef __cplusplus
}
#endif
#endif /* PYNAMA_HIP_H */
