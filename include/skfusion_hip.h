/* skfusion_hip.h -- C ABI of libskfusion_hip.so: the MI355X (gfx950) engine for the DFMF / DFMC /
 * fold-in update loop of scikit-fusion.
 *
 * The reference has no FFI: its seam is the functional solver API that the one-line wrappers
 * call (reference skfusion/fusion/decomposition/dfmf.py:14-15 -> _dfmf.py:127 `dfmf`,
 * dfmc.py:14-15 -> _dfmc.py:181 `dfmc`, dfmf.py:109-115 -> _dfmf.py:330 `transform`).  The entry
 * points below are what a ctypes binding at that seam binds (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success, <0 on error
 *     (SKF_E_*), and the message is available from skf_last_error() (thread-local).
 *   - every data pointer is a DEVICE pointer owned by the caller; the library never allocates
 *     or frees device memory: the caller provides one workspace of skf_plan_workspace_bytes().
 *   - matrices are row-major with a leading dimension counted in elements.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  No call synchronises
 *     the host unless stated.
 *   - a plan is not re-entrant; different plans may be driven from different threads.
 */
#ifndef SKFUSION_HIP_H
#define SKFUSION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* arithmetic type of the engine: storage of the relation / constraint matrices and of every
 * factor; the reference computes in f64 throughout (numpy float64, _init.py:16). */
enum { SKF_F64 = 0, SKF_F32 = 1, SKF_BF16 = 2 /* bf16 relation storage + f32 masters */ };

/* which reference solver the plan replaces */
enum {
    SKF_DFMF = 0,      /* _dfmf.py:127-327  */
    SKF_DFMC = 1,      /* _dfmc.py:181-397  */
    SKF_TRANSFORM = 2  /* _dfmf.py:330-458  */
};

enum { SKF_ENGINE_MFMA = 0, SKF_ENGINE_VALU = 1 };

enum {
    SKF_OK = 0,
    SKF_E_INVALID = -1,    /* bad argument / shape / index */
    SKF_E_STATE = -2,      /* call order (workspace not bound, factors not set ...) */
    SKF_E_WORKSPACE = -3,  /* workspace too small / misaligned */
    SKF_E_HIP = -4         /* a HIP runtime call or kernel launch failed */
};

typedef struct skf_plan skf_plan;

typedef struct {
    int64_t n_obj; /* objects of this type   (count_objects, _dfmf.py:95-124) */
    int32_t rank;  /* factorisation rank c_i (int(ot.rank), dfmf.py:67)       */
} skf_type_desc;

typedef struct {
    int32_t row_type, col_type; /* indices into the type array; row_type != col_type */
    const void* data;           /* n_row x n_col, engine dtype (R[(i,j)][l], dfmf.py:82-85);
                                   SKF_BF16: bf16 (uint16) data, copied ONCE into a zero-padded
                                   row-major layout at bind time (both contractions read that
                                   one copy) and not referenced afterwards */
    int64_t ld;
    const uint8_t* mask;        /* DFMC only (M[(i,j)][l], dfmc.py:77-90); NULL = no mask.
                                   Default: n_row x n_col BYTES, !=0 = unknown entry, mask_ld in
                                   bytes = entries.  SKF_REL_MASK_BITS: packed, one BIT per entry
                                   (bit (c & 7) of byte mask[r * mask_ld + (c >> 3)], 1 = unknown),
                                   mask_ld = bytes per row >= ceil(n_col / 8).  Either form is
                                   converted to the engine's packed layout at bind time and not
                                   referenced afterwards: an iteration reads 1 bit per entry */
    int64_t mask_ld;
    /* Row-block sharding of ONE relation over several processes: this process holds
     * rows [row_begin, row_begin + n_rows) of the relation; `data` / `mask` point at its first LOCAL
     * row.  n_rows == 0 (zero-initialised descriptor): the whole relation is here.  A plan that
     * lists a relation it holds no rows of sets SKF_REL_ABSENT (data may be NULL).  See skf_stage. */
    int64_t row_begin;
    int64_t n_rows;
    int32_t flags;              /* SKF_REL_* */
    int64_t known_bound;        /* masked relations (DFMC): 0 = unknown, or an upper bound on the number of KNOWN entries
                                   (mask == 0).  When that is a small share of the relation, the plan keeps ONLY the known
                                   entries (CSR + CSC with values) and never materialises the completed relation of
                                   _dfmc.py:319-325: with E = (R - G_i S G_j^T) on the known entries, R_c = G_i S G_j^T + E,
                                   so P, Q and the next backbone's G_i^T R_c G_j split into c x c algebra plus products
                                   of the sparse E (cost ~ known * c + n * c^2 instead of n_i * n_j * c; same results up
                                   to associativity).  A bound that turns out too small is an error at bind time.
                                   Ignored for unmasked relations, row blocks and SKF_DFMF / SKF_TRANSFORM plans. */
} skf_relation_desc;

enum {
    SKF_REL_ABSENT = 1,       /* no local rows of this relation (its backbone / Q are still kept) */
    SKF_REL_NO_COL_SIDE = 2,  /* another process adds the column-side terms E_j, D_j of this relation */
    SKF_REL_MASKED = 4,       /* SKF_REL_ABSENT descriptors: the relation is masked where it lives */
    SKF_REL_MASK_BITS = 8,    /* `mask` is packed, one bit per entry (see skf_relation_desc.mask) */
    SKF_REL_KNOWN_LISTS = 32, /* SKF_OPT_OWNED_ROWS plans: this masked relation is kept as lists of its known entries (see
                                 known_bound) on EVERY process -- the caller decides it for all of them alike from the share of
                                 known entries of the WHOLE relation (the partial sums the processes exchange follow one
                                 convention per relation); also set on SKF_REL_ABSENT descriptors of such a relation.  known_bound
                                 then bounds the known entries of the LOCAL rows.  Other plans: ignored (the library decides). */
    SKF_REL_BINARY = 16       /* every entry is 0 or 1 (checked at bind time; "movie has genre", "user tagged").
                                 SKF_BF16 keeps such a relation as a BITMAP -- 1 bit instead of a bf16 per entry
                                 in HBM and on the way to the matrix cores, where it is expanded to bf16 0 / 1 in
                                 LDS: the same products as the dense form, bit for bit.  A sparse one also keeps
                                 the positions of its ones as CSR + CSC and both contractions run over those
                                 lists: with both ranks 64 / 128 / 256 and at most 1 entry in 80 set, as f32 sums
                                 of the bf16 rows of the factors (what the bitmap kernels compute, in another
                                 order); with other ranks and at most 1 entry in 256 set, as gathers of the f32
                                 factor rows (exact f32 sums).  Ignored for masked relations and by the f32 / f64
                                 engines. */
};

typedef struct {
    int32_t type;     /* constrained object type (Theta[(i,i)][t], dfmf.py:82-85) */
    const void* data; /* n_i x n_i, engine dtype (SKF_BF16: f32), dense row-major */
    int64_t ld;
    int64_t nnz;      /* 0: multiply it as a dense matrix, as the reference does (_dfmf.py:284-292).  > 0: an upper
                         bound on its non-zero entries -- the engine then keeps the constraint as CSR (built on the
                         device at bind time, `data` is not referenced afterwards) and D_i += Theta+ G_i,
                         E_i += Theta- G_i become ONE sparse pass in the master precision.  Accepted up to
                         n_i * n_i / 16; a bound that turns out too small is an error at bind time. */
} skf_theta_desc;

typedef struct {
    int32_t dtype;       /* SKF_F64 / SKF_F32 / SKF_BF16 */
    int32_t variant;     /* SKF_DFMF / SKF_DFMC / SKF_TRANSFORM */
    int32_t target_type; /* SKF_TRANSFORM: the type whose factor is folded in */
    int32_t engine;      /* SKF_ENGINE_MFMA (default) / SKF_ENGINE_VALU */
    /* Row-block sharding only (plans whose relations carry row blocks): this process is part
     * `part_index` of `part_count`; it adds the type-level terms E_i += G_i * sum_r B_r^-,
     * D_i += G_i * sum_r B_r^+ (_dfmf.py:260-264,272-276,278-282 summed over the relations, which is
     * linear in B) on its 1/part_count share of the rows of every type.  0, 0 = all rows. */
    int32_t part_index, part_count;
    int32_t flags;       /* SKF_OPT_* (ABI version 4: the field is new; callers check skf_abi_version() when they load the library) */
} skf_options;

enum {
    SKF_OPT_OWNED_ROWS = 1 /* Ownership-aligned row sharding of ONE fit over `part_count` processes (replaces the
                              reference's per-block joblib tasks, _dfmf.py:69-73, _dfmc.py:341-345): process `part_index` OWNS the
                              rows skf_owned_rows() names of EVERY object type -- their factor rows, their E / D accumulators,
                              their update -- and holds exactly those rows of every relation whose row type it is (the
                              descriptors carry that block, or SKF_REL_ABSENT where the process owns no row of the type), and
                              every constraint in full.  Row-side terms, type-level terms and constraint rows then need NO
                              exchange; per iteration a process sends only: the partial Q = R_blk^T G_i[blk] of every relation
                              (reduce-scatter to the owners of the column type), the updated factor rows (all-gather; SKF_BF16:
                              the bf16 operand copy only, unless a constraint on the type reads the f32 rows), and the c x c
                              partial Gram / W matrices (all-reduce).  Such a plan iterates through skf_iterate_dist only. */
};

/* ---- plan life cycle ------------------------------------------------------------------- */

/* Validates the graph (shape consistency is a hard error here; the reference only logs it,
 * _dfmf.py:117-123) and builds the per-iteration launch schedule. */
int skf_plan_create(int32_t n_types, const skf_type_desc* types, int32_t n_relations,
                    const skf_relation_desc* relations, int32_t n_thetas,
                    const skf_theta_desc* thetas, const skf_options* options, skf_plan** out);
int skf_plan_destroy(skf_plan* plan);

int skf_plan_workspace_bytes(const skf_plan* plan, size_t* bytes);
/* `workspace` must be 256-byte aligned device memory and stay valid for the plan's lifetime.
 * For DFMC the masked relations are copied into the workspace here (the caller's relation
 * data is never written: _dfmc.py:268, tests/test_dfmc.py:62,85). */
int skf_plan_bind_workspace(skf_plan* plan, void* workspace, size_t bytes, void* stream);

/* ---- factors --------------------------------------------------------------------------- */

/* Precision model: the n-sized matrices (relations, factors G, accumulators E/D, P, Q) live in
 * the master type -- f64 for SKF_F64, f32 otherwise.  Every c x c matrix (Gram, its
 * pseudo-inverse, backbones S and the B/D terms) is kept and combined in f64 in ALL engines, and
 * the two reductions over the object dimension that feed them (G^T G, G^T P) accumulate in f64:
 * the Gram matrices of real data are ill-conditioned (cond ~1e5 for the reference's dicty
 * example) and an all-f32 pipeline diverges where the f64 reference does not.
 * set_factor copies an n_obj x rank matrix in  set_factor copies an n_obj x rank matrix in
 * (the G0 of `initialize`, _init.py:6-61, or a frozen fitted factor for SKF_TRANSFORM). */
int skf_set_factor(skf_plan* plan, int32_t type, const void* G, int64_t ld, void* stream);
int skf_get_factor(const skf_plan* plan, int32_t type, void* G, int64_t ld, void* stream);
/* backbone S of relation `rel` (rank_row x rank_col).  set: SKF_TRANSFORM only (frozen S). */
int skf_set_backbone(skf_plan* plan, int32_t rel, const void* S, int64_t ld, void* stream);
int skf_get_backbone(const skf_plan* plan, int32_t rel, void* S, int64_t ld, void* stream);

/* ---- the hot path ---------------------------------------------------------------------- */

/* Runs `n_iters` full update iterations (the body of `for iter in range(max_iter)`,
 * _dfmf.py:212-296 / _dfmc.py:270-366 / _dfmf.py:370-428) device-resident on `stream`.
 * After the call get_backbone returns the S computed from the factors BEFORE the last G update
 * and get_factor the factors AFTER it -- the same generation mismatch the reference returns
 * (_dfmf.py:239 vs :295, return :327). */
int skf_iterate(skf_plan* plan, int32_t n_iters, void* stream);
/* The same for `n_plans` (1 .. 64) plans of ONE graph at once -- independent random restarts (the reference's n_run loop
 * over joblib workers, dfmf.py:87-95) of a graph too small to fill the chip: every launch of the iteration serves all of
 * them (the restart is a grid dimension), so ten restarts of the dicty graph cost little more than one.  The plans must
 * have been created from the same object types, ranks, relation shapes and engine and must run the schedule for small
 * graphs (SKF_DFMF, every rank <= 64, sparse constraints, at most 8192 objects per type); each keeps its own workspace,
 * factors and results, exactly as after skf_iterate.  SKF_E_STATE when the plans do not batch: iterate them one by one.
 * SKF_TRANSFORM plans batch as well: the fold-ins of ONE set of new relations into the models of several restarts (the
 * reference's n_run fold-ins over joblib workers, dfmf.py:191-199) -- same object types, ranks and relations, each plan with
 * the frozen factors / backbones of its restart, no constraint on the target type; one launch per iteration serves all. */
int skf_iterate_batch(skf_plan* const* plans, int32_t n_plans, int32_t n_iters, void* stream);
/* *yes = 1 when the (bound) plan can be one of the plans of skf_iterate_batch: the schedule for small graphs, or a fold-in
 * without constraints on the target type. */
int skf_plan_batchable(const skf_plan* plan, int32_t* yes);
/* The limits behind that verdict for SKF_DFMF plans in f32 / f64 (the three-launch schedule of small graphs): every rank
 * <= *max_rank, every type <= *max_objects objects, at most *max_types / *max_relations / *max_constraints of each, every
 * constraint sparse (0 < nnz <= n * n / *constraint_nnz_divisor), no masked, absent or row-block relation -- so that a host
 * layer can decide whether restarts will share launches WITHOUT uploading the graph (the reference hands restarts to joblib
 * workers, dfmf.py:87-95; here the host picks between skf_iterate_batch and one plan after the other).  Null pointers are
 * skipped. */
int skf_small_graph_limits(int32_t* max_rank, int64_t* max_objects, int32_t* max_types, int32_t* max_relations,
                           int32_t* max_constraints, int32_t* constraint_nnz_divisor);
/* enable != 0: iterations 2..n of skf_iterate replay ONE captured hipGraph (a single host call per
 * iteration instead of one per kernel).  Off by default -- a single fit is bound by kernel time --
 * and switched on for restarts that run CONCURRENTLY on several streams of one GPU, where the
 * host-side launch rate of one thread is the limit (the reference's n_jobs over n_run restarts,
 * dfmf.py:87-95).  A failed capture falls back to eager launches. */
int skf_plan_set_graph(skf_plan* plan, int32_t enable);

/* The same iteration in two halves, for runs whose relations are partitioned over several GPUs
 * (one process and one plan per GPU, each plan holding ALL object types but only its share of
 * the relations / constraints; factors replicated):
 *     skf_accumulate   : Gram, pinv, contractions, backbones and the E / D accumulator sums of
 *                        the plan's own relations and constraints
 *     <all-reduce(sum) of the accumulator range over the ranks -- RCCL via torch.distributed>
 *     skf_apply_update : G <- G * sqrt(E / max(D, eps))          (identical on every rank)
 * skf_accumulator_range reports the byte range [offset, offset+bytes) of the workspace that
 * holds every E and D matrix (master dtype; padding between slots is zero-safe to reduce). */
int skf_accumulate(skf_plan* plan, void* stream);
int skf_apply_update(skf_plan* plan, void* stream);
int skf_accumulator_range(const skf_plan* plan, size_t* offset, size_t* bytes);

/* Row-block sharding: every process creates a plan with ALL object types and ALL relations, each
 * relation carrying this process's row block (or SKF_REL_ABSENT); exactly one holder per relation
 * leaves SKF_REL_NO_COL_SIDE clear.  One iteration is then four stages with an all-reduce(sum)
 * over the processes of a workspace byte range between them (skf_exchange_range; the ranges have
 * the same layout in every such plan):
 *     skf_stage(SKF_STAGE_CONTRACT)    Gram, pinv, local P = R_blk G_j, partial Q = R_blk^T G_i[blk],
 *                                      partial W = G_i[blk]^T P          -> reduce SKF_X_W, SKF_X_Q
 *     skf_stage(SKF_STAGE_BACKBONE)    S = K_i W K_j; DFMC: completion of the local rows, P and
 *                                      partial Q of masked relations     -> reduce SKF_X_QM (DFMC)
 *     skf_stage(SKF_STAGE_ACCUMULATE)  E / D terms of the local rows, column-side and Theta terms
 *                                                                        -> reduce SKF_X_ED
 *     skf_stage(SKF_STAGE_UPDATE)      G <- G * sqrt(E / max(D, eps))
 * (the +- split of _dfmf.py:256-276 is non-linear, hence Q is reduced raw, before it).
 * skf_accumulate == the first three stages back to back (plans without row blocks);
 * skf_iterate refuses a plan with row blocks.  skf_relation_sqerr covers the local rows. */
enum { SKF_STAGE_CONTRACT = 0, SKF_STAGE_BACKBONE = 1, SKF_STAGE_ACCUMULATE = 2, SKF_STAGE_UPDATE = 3 };
enum { SKF_X_W = 0, SKF_X_Q = 1, SKF_X_QM = 2, SKF_X_ED = 3 };
int skf_stage(skf_plan* plan, int32_t stage, void* stream);
/* dtype: SKF_F64 for SKF_X_W, the master type otherwise; bytes may be 0 (nothing to reduce). */
int skf_exchange_range(const skf_plan* plan, int32_t which, size_t* offset, size_t* bytes, int32_t* dtype);

/* ---- collectives behind the boundary ------------------------------------------------------------------------------
 * The reference spreads its restarts over joblib workers (dfmf.py:87-95) and its block products over `_par_bdot`
 * (_dfmf.py:69-73); a fit spread over several GPUs needs the sums of the E / D accumulators (and, with row blocks, of W
 * and Q) instead.  A communicator is either RCCL -- bound with dlopen at run time (the librccl of the process if one is
 * loaded, else SKF_RCCL_PATH / the loader path / /opt/rocm/lib), one process per GPU, the current device at
 * skf_comm_create -- or a callback the caller supplies (another transport; the CPU tests run it over gloo).
 *     rank 0: skf_comm_unique_id(id)  -> hand the 128 bytes to every rank ->  skf_comm_create(id, rank, world, &comm)
 *     skf_plan_set_comm(plan, comm);  skf_iterate_dist(plan, n_iters, stream)
 * skf_iterate_dist = the same iteration as skf_accumulate + all-reduce + skf_apply_update (plans without row blocks) or the
 * four skf_stage calls with their exchanges (plans with row blocks), issued on `stream` by the library itself: W, Q as
 * all-reduce; E and D as REDUCE-SCATTER over `world` equal element ranges of the accumulator regions, the multiplicative
 * update of the owned range of G, and an ALL-GATHER of G -- 3 (world-1)/world of the factor bytes per rank and iteration
 * instead of the 4 (world-1)/world of all-reducing both accumulators (config 3 on 8 GPUs: 444 MB instead of 592 MB).  Every
 * rank ends an iteration with identical factors.  skf_exchange_bytes: bytes one rank sends per iteration on a ring.
 * SKF_OPT_OWNED_ROWS plans (ownership-aligned row blocks): no exchange of E / D at all -- all-reduce of the c x c partial
 * Gram and W matrices, one reduce-scatter per relation of its partial Q to the owners of the column type, update of the
 * OWNED rows, one all-gather per type of the updated rows (SKF_BF16: of their bf16 operand copy; the f32 rows of the other
 * owners are fetched once, at the end of the call).  The exchanges go out on a stream of their own as soon as their input
 * is complete -- a relation's Q under the next relation's contractions, a type's gather as soon as its last relation is
 * through -- and the communicator's (rank, world) must be the plan's (part_index, part_count).  Config 3 on 8 GPUs, bf16:
 * 172 MB per rank and iteration instead of 605 MB. */
typedef struct skf_comm skf_comm;
/* op 0: all-reduce(sum) of `count` elements in place; 1: reduce-scatter(sum) -- `buf` holds world * count elements, on
 * return elements [rank * count, (rank + 1) * count) are reduced; 2: all-gather of that range to every rank, in place.
 * dtype SKF_F64 / SKF_F32 (SKF_BF16: 2-byte elements, all-gather only); `buf` is device memory, the call is ordered on
 * `stream`; return 0 on success. */
typedef int (*skf_collective_fn)(void* user, int32_t op, void* buf, size_t count, int32_t dtype, void* stream);
int skf_comm_unique_id(void* id128);
int skf_comm_create(const void* id128, int32_t rank, int32_t world, skf_comm** out);   /* id128 == NULL: world must be 1 */
int skf_comm_create_callback(int32_t rank, int32_t world, skf_collective_fn fn, void* user, skf_comm** out);
/* A communicator that exchanges nothing: `rank` of `world` on ONE device, for timing the compute of that rank of a
 * sharded fit where the other ranks do not exist (bench.py --emulate-rank k/W).  A sum over the ranks is stood in for by
 * world x this rank's partial sum and the multiplicative update of the owned rows is not applied (so that the Gram matrices
 * the timed launches see stay those of well-conditioned factors and the pseudo-inverses take the path they take in a real
 * run); gathers leave the other ranks' rows as they were.  Every other launch of the iteration runs. */
int skf_comm_create_null(int32_t rank, int32_t world, skf_comm** out);
/* What a communicator is: its rank / world as created, its transport, and how many ranks the transport itself reports
 * (SKF_COMM_RCCL: ncclCommCount of the bound communicator, -1 when the library does not export it; callback: world;
 * single: 1; null: 0) -- lets a caller (bench.py's strong-scaling record) state that RCCL really spans the ranks of the
 * run.  Null pointers are skipped. */
enum { SKF_COMM_SINGLE = 0, SKF_COMM_RCCL = 1, SKF_COMM_CALLBACK = 2, SKF_COMM_NULL = 3 };
int skf_comm_info(const skf_comm* comm, int32_t* rank, int32_t* world, int32_t* transport, int32_t* transport_ranks);
int skf_comm_destroy(skf_comm* comm);
int skf_plan_set_comm(skf_plan* plan, skf_comm* comm);     /* not owned by the plan; NULL detaches */
int skf_iterate_dist(skf_plan* plan, int32_t n_iters, void* stream);
int skf_exchange_bytes(const skf_plan* plan, int32_t world, size_t* bytes);

/* sum over the relation of (R - G_i S G_j^T)^2 with the current (G, S), written as one f64 to
 * the DEVICE address `out` (reconstruction error of _dfmf.py:306-316 without materialising the
 * n_i x n_j product).  For DFMC the working copy (completed entries) is used.  SKF_BF16: one pass
 * over the stored bf16 relation, bf16-rounded G_i S and G_j on the matrix cores, f32 residual. */
int skf_relation_sqerr(skf_plan* plan, int32_t rel, double* out, void* stream);

/* The two contraction results the LAST iteration left in the workspace, for verification at sizes where the
 * host cannot recompute them: which = 0: P = R G_j (local rows x rank_col), 1: Q = R^T G_i (n_col x rank_row),
 * both from the factors BEFORE that iteration's update (like the backbone), master dtype, copied to `dst`.
 * A masked relation kept as known entries only (skf_relation_desc.known_bound) never forms P: it answers
 * which = 2 with the row-side product P S^T (n_row x rank_row) it computes instead, and which = 1 as usual. */
int skf_get_contraction(const skf_plan* plan, int32_t rel, int32_t which, void* dst, int64_t ld, void* stream);

/* Optional hipEvent timing of the launches that walk a relation (P = R G_j, Q = R^T G_i -- the dominant
 * kernel -- and their sparse counterparts).  get_profile synchronises on the recorded events, returns the summed
 * duration [ms], the number of launches, the flops they EXECUTE (2*M*N*K for a product on the matrix cores, dense or
 * bitmap; 2 * ones * N for the row gathers of a very sparse 0/1 relation; 2 or 4 * known * c for a pass over the
 * known entries of a masked relation) and the relation bytes they read from HBM as stored (bf16 / f32 / f64 entries,
 * 1 bit per entry for a bitmap, index + value lists for the sparse forms) since the last call, and resets the counters. */
int skf_plan_set_profiling(skf_plan* plan, int32_t enable);
int skf_plan_get_profile(skf_plan* plan, double* total_ms, int64_t* launches, double* flops, double* bytes);

/* ---- stand-alone operators (building blocks, exported for tests / callers) -------------- */

typedef struct {
    const void* A; /* A(m,k) = A[m*sa_m + k*sa_k] */
    const void* B; /* B(k,n) = B[k*sb_k + n*sb_n] */
    void* C;
    void* C2;
    const uint8_t* mask;
    int64_t sa_m, sa_k, sb_k, sb_n, ldc, ldc2, ldmask;
    int32_t M, N, K;
    int32_t aop;        /* 0: x   1: max(x,0)   2: max(-x,0)  applied to A on load */
    int32_t epi;        /* 0 store, 1 accumulate, 2 split-store, 3 split-accumulate, 4 masked store */
    int32_t nan_to_num; /* numpy.nan_to_num on the product before the epilogue */
    int32_t splits;     /* 0 = choose; >1 needs workspace of splits*M*N elements */
    int32_t a_dtype;    /* storage type of A / B: -1 = same as `dtype`, else SKF_F64 / SKF_F32.   */
    int32_t b_dtype;    /* supported (C,A,B): (f64,f64,f64) (f32,f32,f32) (f64,f32,f32) (f32,f32,f64) */
} skf_gemm_desc;

/* C = epi(aop(A) * B) on the matrix cores; `dtype` = arithmetic and result type (f64 / f32). */
int skf_gemm(int32_t dtype, int32_t engine, const skf_gemm_desc* desc, void* workspace,
             size_t workspace_bytes, void* stream);

/* bf16 relation contraction C[M x N] (f32) = A[M x Kp] * Bt[N x Kp]^T, both operands bf16,
 * row-major and K-contiguous, Kp a multiple of 64 with zero-filled padding, lda/ldb multiples of
 * 8 and 16-byte aligned bases (v_mfma_f32_16x16x32_bf16, f32 accumulation).  `splits` = 0 lets
 * the library choose the K slicing; >1 needs a workspace of splits*M*N floats. */
int skf_gemm_bf16(const void* A, int64_t lda, const void* Bt, int64_t ldb, float* C, int64_t ldc,
                  int32_t M, int32_t N, int32_t Kp, int32_t splits, void* workspace,
                  size_t workspace_bytes, void* stream);

/* The transposed-A form of the same contraction, C[M x N] (f32) = A[Kp x M]^T * Bt[N x Kp]^T: A is
 * row-major [Kp][lda] with the OUTPUT rows along its columns (lda >= M, lda % 8 == 0, rows zero-filled
 * up to Kp % 64 == 0).  This is Q = R^T G_i read from the row-major relation itself: the A tile lands in
 * LDS as stored and the fragments are read transposed (ds_read_b64_tr_b16) -- no stored R^T
 * (reference _dfmf.py:266 multiplies the strided view R.T the same way, without a copy). */
int skf_gemm_bf16_tn(const void* A, int64_t lda, const void* Bt, int64_t ldb, float* C, int64_t ldc,
                     int32_t M, int32_t N, int32_t Kp, int32_t splits, void* workspace,
                     size_t workspace_bytes, void* stream);

/* The same two contractions with a BINARY A operand stored as a bitmap (SKF_REL_BINARY): bit (c & 7) of byte
 * A[r * lda_bytes + (c >> 3)] is entry (r, c), lda_bytes a multiple of 8, padding bits zero.
 * transposed == 0: C[M x N] = A[M x Kp] * Bt^T (rows of the bitmap = output rows);
 * transposed != 0: C[M x N] = A[Kp x M]^T * Bt^T (rows of the bitmap = the contraction index, zero rows up to Kp). */
int skf_gemm_bits(const void* A, int64_t lda_bytes, const void* Bt, int64_t ldb, float* C, int64_t ldc,
                  int32_t M, int32_t N, int32_t Kp, int32_t transposed, int32_t splits, void* workspace,
                  size_t workspace_bytes, void* stream);

/* dst (bf16, ld ldd) = round-to-nearest-even(src) or its transpose; src dtype SKF_F64 / SKF_F32 /
 * SKF_BF16.  Padding columns of dst are left untouched (zero them beforehand). */
int skf_to_bf16(void* dst, int64_t ldd, int32_t src_dtype, const void* src, int64_t lds,
                int64_t rows, int64_t cols, int32_t transpose, void* stream);

/* K = pinv(A) for symmetric A (n x n): f64 Jacobi eigen-decomposition + the singular-value
 * cut-off of scipy.linalg.pinv (reference _dfmf.py:232, _dfmc.py:307). */
int skf_pinv_sym_workspace_bytes(int32_t n, size_t* bytes);
int skf_pinv_sym(int32_t dtype, const void* A, int64_t lda, void* K, int64_t ldk, int32_t n,
                 void* workspace, size_t workspace_bytes, void* stream);

/* dst(r,c) = hash_uniform(seed, r*cols + c) * scale + shift  (synthetic benchmark data) */
int skf_fill_uniform(int32_t dtype, void* dst, int64_t rows, int64_t cols, int64_t ld,
                     uint64_t seed, double scale, double shift, void* stream);

/* Relation.filled() on the device (reference fusion_graph.py:464-510): every UNKNOWN entry of `data` (rows x cols,
 * SKF_F64 / SKF_F32, in place) -- not finite, or flagged in the optional byte mask -- is replaced by the mean of the
 * known entries of the whole matrix (strategy 0, 'mean'), of its row (1, 'row_mean') or column (2, 'col_mean'; rows /
 * columns without a known entry take the overall mean), or by `value` (3, a constant).  numpy.nanmean semantics: NaN
 * and masked entries are skipped, +-inf take part in the means.  Whether the mask survives the fill is the host
 * layer's business (it does for 'mean' and constants, not for the row / column means: fusion_graph.py:475-489). */
int skf_fill_unknown_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes);
int skf_fill_unknown(int32_t dtype, void* data, int64_t ld, int64_t rows, int64_t cols, const uint8_t* mask,
                     int64_t mask_ld, int32_t strategy, double value, void* workspace, size_t workspace_bytes,
                     void* stream);

/* dst(r,c) = (dst_dtype) src(r,c), dtypes SKF_F64 / SKF_F32 */
int skf_cast(int32_t dst_dtype, void* dst, int64_t ldd, int32_t src_dtype, const void* src,
             int64_t lds, int64_t rows, int64_t cols, void* stream);

/* Rows [*begin, *begin + *count) of a type of `n_obj` objects that process `part_index` of `part_count` owns under
 * SKF_OPT_OWNED_ROWS; *chunk = rows per process of the padded layout the exchanges use (equal for all processes, a multiple
 * of 256 for large types / of 64 for SKF_BF16; part_count * chunk >= n_obj, the last owners may hold fewer rows or none). */
int skf_owned_rows(int32_t dtype, int64_t n_obj, int32_t part_index, int32_t part_count, int64_t* begin, int64_t* count,
                   int64_t* chunk);

/* Kernel launches the calling thread has issued through this library so far (every plan, every entry point; a hipGraph
 * replay counts the launches it was captured from once, at capture).  Callers take differences: launches per iteration of a
 * schedule are part of its latency budget (bench.py reports them for the rank-of-8 emulation). */
int skf_launch_count(int64_t* launches);

/* Split-K launches of the calling thread whose slice count was CLAMPED because the scratch sized at plan creation could not
 * hold the slices the launch-time model asked for (run_gemm: safe, but the modelled schedule is then not the executed one;
 * advisor, round 5).  Zero for every plan the tests and the benchmark create; a non-zero count names a sizing rule
 * (skf_plan_create: want_part) that has fallen behind the tile / slice pickers.  The stand-alone products (skf_gemm*)
 * run on the caller's scratch and are not counted. */
int skf_split_clamps(int64_t* clamps);

const char* skf_last_error(void);
const char* skf_version(void);
/* Layout version of the structs and signatures above (SKF_ABI_VERSION).  A binding built against another version must not
 * call the library: descriptors grew between versions (skf_relation_desc.known_bound: 3, skf_options.flags: 4; version 5
 * adds entry points only -- skf_small_graph_limits, skf_comm_info, skf_launch_count -- the structs are those of version 4). */
#define SKF_ABI_VERSION 5
int skf_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SKFUSION_HIP_H */
