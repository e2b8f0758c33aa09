// skf_api.hip -- C ABI (include/skfusion_hip.h) and the host-side plan / launch schedule of the
// DFMF / DFMC / fold-in iteration.  Kernels: skf_kernels.h.
//
// One iteration (reference _dfmf.py:228-296, 2-GEMM form of SURVEY.md 7.0; everything reads the
// OLD factors, G is replaced at the very end):
//   Gram_i = G_i^T G_i                    split-K MFMA GEMM + fixed-order reduce   (:228-231)
//   K_i    = pinv(Gram_i)                 Jacobi eigen kernel, one workgroup/type  (:232)
//   P_r = R_r G_j ; Q_r = R_r^T G_i       the two big contractions (only reads of R)
//   S_r = K_i (G_i^T P_r) K_j             (:236-239)
//   [DFMC] R_r[mask] = (G_i S_r G_j^T)[mask], P_r recomputed               (_dfmc.py:319-325)
//   E_i += (P_r S_r^T)+ + G_i B-  ; D_i += (P_r S_r^T)- + G_i B+ ,  B = S Gram_j S^T  (:254-281)
//   E_j += (Q_r S_r)+   + G_j D-  ; D_j += (Q_r S_r)-   + G_j D+ ,  D = S^T Gram_i S
//   D_i += Theta+ G_i ; E_i += Theta- G_i                                   (:284-292)
//   G_i <- G_i * sqrt(E_i / max(D_i, eps))                                  (:294-296)
#include "skf_kernels.h"
#include "skf_known.h"
#include "skf_small.h"
// the product build compiles the GEMM-class kernel templates in units of their own, side by side with this file
// (tools/gen_inst_units.py, __graft_entry__.build); a build of this file alone (emulator, probes) instantiates them here
#ifdef SKF_SPLIT_BUILD
#include "skf_inst_decl.h"
#endif

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/skfusion_hip.h"

struct skf_comm;

namespace skf {

static thread_local std::string g_err;

struct Error {
    int code;
    std::string msg;
};

#define SKF_FAIL(code_, ...)                                  \
    do {                                                      \
        char buf_[512];                                       \
        snprintf(buf_, sizeof buf_, __VA_ARGS__);             \
        throw Error{(code_), std::string(buf_)};              \
    } while (0)

#define SKF_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) SKF_FAIL(SKF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// the ONE place the library reads its environment (test / A-B switches: Switches::read and plan creation; SKF_RCCL_PATH)
static const char* env_str(const char* name) { return getenv(name); }
static int env_int(const char* name, int unset) {
    const char* v = env_str(name);
    return v ? atoi(v) : unset;
}

static thread_local int64_t g_split_clamps = 0;  // split-K launches whose slice count the scratch could not hold (skf_split_clamps)
static thread_local int64_t g_launches = 0;      // kernel launches this thread has issued through the library (skf_launch_count)

static inline void check_launch(const char* what) {
    ++g_launches;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) SKF_FAIL(SKF_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
}

template <class F>
static int guarded(F&& f) {
    try {
        f();
        return SKF_OK;
    } catch (const Error& e) {
        g_err = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        g_err = e.what();
        return SKF_E_INVALID;
    }
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// SKF_OPT_OWNED_ROWS: rows per rank of a type of n objects over `world` ranks (include/skfusion_hip.h, skf_owned_rows):
// boundaries at multiples of the 256-row contraction tile for large types, of 64 for SKF_BF16 (the K padding of Q = R^T G_i)
static inline int64_t owned_chunk(int dtype, int64_t n, int world) {
    const int64_t per = (n + world - 1) / world;
    const int64_t a = per >= 1024 ? 256 : (dtype == 2 /* SKF_BF16 */ ? 64 : 1);
    return (per + a - 1) / a * a;
}
static inline int elem_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;           // grid-stride the rest (256 CUs x 8 blocks)
    return (int)b;
}

#include "skf_gemm_launch.inc"

#include "skf_plan.inc"

#include "skf_stages.inc"

#include "skf_dist.inc"

#include "skf_schedule.inc"

}  // namespace skf

using namespace skf;

extern "C" {

const char* skf_last_error(void) { return g_err.c_str(); }
const char* skf_version(void) { return "skfusion_hip 0.1 (gfx950)"; }

int skf_plan_create(int32_t n_types, const skf_type_desc* types, int32_t n_relations,
                    const skf_relation_desc* relations, int32_t n_thetas, const skf_theta_desc* thetas,
                    const skf_options* opt, skf_plan** out) {
    return guarded([&] {
        if (!out || !types || !opt || n_types <= 0) SKF_FAIL(SKF_E_INVALID, "null argument / no object types");
        if (n_relations < 0 || n_thetas < 0 || (n_relations > 0 && !relations) || (n_thetas > 0 && !thetas))
            SKF_FAIL(SKF_E_INVALID, "bad relation / constraint arrays");
        if (opt->dtype != SKF_F64 && opt->dtype != SKF_F32 && opt->dtype != SKF_BF16)
            SKF_FAIL(SKF_E_INVALID, "unknown dtype %d", opt->dtype);
        if (opt->dtype == SKF_BF16 && opt->engine != SKF_ENGINE_MFMA)
            SKF_FAIL(SKF_E_INVALID, "SKF_BF16 needs the MFMA engine");
        if (opt->variant < SKF_DFMF || opt->variant > SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "bad variant");
        if (opt->engine != SKF_ENGINE_MFMA && opt->engine != SKF_ENGINE_VALU) SKF_FAIL(SKF_E_INVALID, "bad engine");
        skf_plan* p = new skf_plan();
        struct Guard { skf_plan* p; ~Guard() { delete p; } } guard{p};
        p->dtype = opt->dtype;
        p->variant = opt->variant;
        p->engine = opt->engine;
        p->f64 = (opt->dtype == SKF_F64);
        p->bf16 = (opt->dtype == SKF_BF16);
        p->esz = p->f64 ? 8 : 4;
        p->mt = p->f64 ? SKF_F64 : SKF_F32;
        p->target = opt->target_type;
        if (p->variant == SKF_TRANSFORM && (p->target < 0 || p->target >= n_types))
            SKF_FAIL(SKF_E_INVALID, "target type %d out of range", p->target);
        p->types.resize(n_types);
        for (int i = 0; i < n_types; ++i) {
            if (types[i].n_obj <= 0 || types[i].rank <= 0 || types[i].rank > EIGH_MAXN - 1)
                SKF_FAIL(SKF_E_INVALID, "object type %d: n_obj=%lld rank=%d invalid", i, (long long)types[i].n_obj,
                         types[i].rank);
            if (types[i].n_obj > 2000000000LL) SKF_FAIL(SKF_E_INVALID, "object type %d too large", i);
            p->types[i].n = types[i].n_obj;
            p->types[i].c = types[i].rank;
            p->types[i].n_pad = (types[i].rank + 1) / 2 * 2;
            p->types[i].t0 = 0;
            p->types[i].tn = types[i].n_obj;
            p->types[i].n_alloc = types[i].n_obj;
            if (opt->flags & SKF_OPT_OWNED_ROWS) {      // the rows this process owns (skf_owned_rows)
                if (opt->part_count < 1 || opt->part_index < 0 || opt->part_index >= opt->part_count)
                    SKF_FAIL(SKF_E_INVALID, "SKF_OPT_OWNED_ROWS: part_index %d outside [0, %d)", opt->part_index, opt->part_count);
                const int64_t ch = owned_chunk(opt->dtype, types[i].n_obj, opt->part_count);
                int64_t lo = ch * opt->part_index, hi = lo + ch;
                if (lo > types[i].n_obj) lo = types[i].n_obj;
                if (hi > types[i].n_obj) hi = types[i].n_obj;
                p->types[i].t0 = lo;
                p->types[i].tn = hi - lo;
                p->types[i].chunk = ch;
                p->types[i].n_alloc = ch * opt->part_count;
            } else if (opt->part_count > 1) {      // even shares of the rows, boundaries at multiples of 64
                if (opt->part_index < 0 || opt->part_index >= opt->part_count)
                    SKF_FAIL(SKF_E_INVALID, "part_index %d outside [0, %d)", opt->part_index, opt->part_count);
                const int64_t per = (types[i].n_obj + opt->part_count - 1) / opt->part_count;
                const int64_t step = (per + 63) / 64 * 64;
                int64_t lo = step * opt->part_index, hi = lo + step;
                if (lo > types[i].n_obj) lo = types[i].n_obj;
                if (hi > types[i].n_obj) hi = types[i].n_obj;
                p->types[i].t0 = lo;
                p->types[i].tn = hi - lo;
            }
        }
        if (opt->part_count > 1) p->sliced = true;
        if (opt->flags & SKF_OPT_OWNED_ROWS) {
            if (p->variant == SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "SKF_OPT_OWNED_ROWS is for SKF_DFMF / SKF_DFMC plans");
            p->owned = p->sliced = true;
            p->part_index = opt->part_index;
            p->part_count = opt->part_count;
        }
        p->rels.resize(n_relations);
        for (int r = 0; r < n_relations; ++r) {
            const skf_relation_desc& d = relations[r];
            if (d.row_type < 0 || d.row_type >= n_types || d.col_type < 0 || d.col_type >= n_types)
                SKF_FAIL(SKF_E_INVALID, "relation %d: type index out of range", r);
            if (d.row_type == d.col_type) SKF_FAIL(SKF_E_INVALID, "relation %d: row type == column type (pass it as a constraint)", r);
            const bool absent = (d.flags & SKF_REL_ABSENT) != 0;
            if (!absent && (!d.data || d.ld < p->types[d.col_type].n))
                SKF_FAIL(SKF_E_INVALID, "relation %d: dimension mismatch (ld %lld < %lld columns)", r, (long long)d.ld,
                         (long long)p->types[d.col_type].n);
            const int64_t n_row_type = p->types[d.row_type].n;
            if (d.row_begin < 0 || d.n_rows < 0 || d.row_begin + d.n_rows > n_row_type)
                SKF_FAIL(SKF_E_INVALID, "relation %d: row block [%lld, +%lld) outside the %lld objects of its row type",
                         r, (long long)d.row_begin, (long long)d.n_rows, (long long)n_row_type);
            const bool block = absent || (d.n_rows > 0 && d.n_rows < n_row_type) || (d.flags & SKF_REL_NO_COL_SIDE);
            if (block && p->variant == SKF_TRANSFORM)
                SKF_FAIL(SKF_E_INVALID, "relation %d: row blocks are for SKF_DFMF / SKF_DFMC plans", r);
            if (block && p->bf16 && d.row_begin % 64 != 0)
                SKF_FAIL(SKF_E_INVALID, "relation %d: SKF_BF16 row blocks must start at a multiple of 64", r);
            if (block) p->sliced = true;
            if ((d.mask || (d.flags & SKF_REL_MASKED)) && p->variant != SKF_DFMC)
                SKF_FAIL(SKF_E_INVALID, "relation %d: masks need SKF_DFMC", r);
            if (d.mask && d.mask_ld < ((d.flags & SKF_REL_MASK_BITS) ? (p->types[d.col_type].n + 7) / 8 : p->types[d.col_type].n))
                SKF_FAIL(SKF_E_INVALID, "relation %d: mask ld", r);
            if (p->variant == SKF_TRANSFORM && d.row_type != p->target && d.col_type != p->target)
                SKF_FAIL(SKF_E_INVALID, "relation %d must include the target object type", r);
            RelState& s = p->rels[r];
            s.row = d.row_type; s.col = d.col_type;
            s.R_in = d.data; s.ld_in = d.ld; s.mask = d.mask; s.ldmask = d.mask_ld;
            s.mask_is_bits = (d.flags & SKF_REL_MASK_BITS) != 0;
            s.binary = p->bf16 && (d.flags & SKF_REL_BINARY) != 0 && !d.mask && !absent;
            s.R = d.data; s.ldr = d.ld;
            s.absent = absent;
            s.r0 = absent ? 0 : d.row_begin;
            s.nr = absent ? 0 : (d.n_rows > 0 ? d.n_rows : n_row_type - d.row_begin);
            s.col_side = (d.flags & SKF_REL_NO_COL_SIDE) == 0;
            s.masked = d.mask != nullptr || (d.flags & SKF_REL_MASKED) != 0;
            if (absent) { s.R_in = s.R = nullptr; s.mask = nullptr; }
            if (p->owned) {         // the block of a relation is the owned range of its row type, nothing else
                const TypeState& ti = p->types[d.row_type];
                if (ti.tn == 0 ? !absent : (absent || s.r0 != ti.t0 || s.nr != ti.tn))
                    SKF_FAIL(SKF_E_INVALID, "relation %d: SKF_OPT_OWNED_ROWS wants the rows [%lld, +%lld) of its row type here (skf_owned_rows)",
                             r, (long long)ti.t0, (long long)ti.tn);
                s.col_side = true;  // every process adds the column-side terms of ITS rows of the column type
            }
            if (d.known_bound < 0) SKF_FAIL(SKF_E_INVALID, "relation %d: negative bound on the known entries", r);
            s.kn_cap = (s.mask && p->variant == SKF_DFMC) ? d.known_bound : 0;
        }
        // masked relations with few known entries are kept as lists of those entries (skf_known.h).  The three passes over the
        // lists gather rank_row-wide vectors -- ~70 ps per entry at rank 128 against ~2.2 ps per CELL for the four passes of
        // the dense path over the completed relation (config 5, bf16) -- hence: known share * rank_row <= 4 (1/32 at rank
        // 128).  SKF_DFMC_SPARSE=0: never; =1: whenever a bound is given (up to a quarter of the relation).  Plans with row
        // blocks keep the dense form.
        {
            const Switches sw0 = Switches::read();          // (plan creation: the plan's own copy is read when its workspace is bound)
            const int mode = sw0.dfmc_sparse;
            // Parts of the lists (skf_known.h): with srp_bf16_v6_kernel the passes are no longer bound by instruction issue and
            // pinning slices of the gathered matrix to XCDs pays (profiles/r03_srp_v6.txt: 25.6 MB of user factors, 8 parts:
            // 1.75 -> 1.10 ms; 10 MB, 4 parts: 1.41 -> 1.25 ms) -- the smallest power of two that brings a slice under the
            // 4 MiB L2 of an XCD, as long as a segment still holds a batch of entries.  Other engines / widths: 1 (their
            // kernels are issue-bound; measured neutral in round 3).  SKF_KNOWN_PARTS=1|2|4|8 overrides.
            const int parts_env = sw0.known_parts;
            auto pick_parts = [&](int64_t n_in, int64_t n_out, int ci, int64_t cap) {
                if (parts_env) return parts_env;
                if (!p->bf16 || (ci != 128 && ci != 256)) return 1;
                int q = 1;
                while (q < 8 && (double)n_in * ci * 2.0 / q > 3.5 * 1048576.0) q *= 2;
                while (q > 1 && (double)cap / ((double)n_out * q) < 64.0) q /= 2;
                return q;
            };
            for (size_t rk = 0; rk < p->rels.size(); ++rk) {
                RelState& s = p->rels[rk];
                const int ci = p->types[s.row].c;
                if (p->owned) {
                    // ownership-aligned row blocks: the caller decided for ALL processes alike (SKF_REL_KNOWN_LISTS); a process
                    // without rows of the relation keeps the flag -- it adds the dense part of Q on ITS rows of the column type
                    const bool lists = (relations[rk].flags & SKF_REL_KNOWN_LISTS) != 0 && s.masked && p->variant == SKF_DFMC;
                    if (lists && (ci > 64 * SRP_MAXREP || s.kn_cap > 2000000000LL || (!s.absent && s.kn_cap <= 0)))
                        SKF_FAIL(SKF_E_INVALID, "relation %zu: SKF_REL_KNOWN_LISTS needs a bound on the known entries of the local rows", rk);
                    if (!lists) {
                        s.kn_cap = 0;
                        continue;
                    }
                } else {
                    if (s.kn_cap <= 0) continue;
                    const double cells = (double)s.nr * (double)p->types[s.col].n;
                    const double share = cells > 0 ? (double)s.kn_cap / cells : 1.0;
                    if (p->sliced || mode == 0 || share > 0.25 || (mode != 1 && share * ci > 4.0) || ci > 64 * SRP_MAXREP ||
                        s.kn_cap > 2000000000LL) {
                        s.kn_cap = 0;
                        continue;
                    }
                }
                s.kn = true;
                if (s.absent) continue;             // (no lists here: only the flag)
                s.kn_pc = pick_parts(p->types[s.col].n, s.nr, ci, s.kn_cap);        // row lists gather the column objects' vectors
                s.kn_pr = pick_parts(s.nr, p->types[s.col].n, ci, s.kn_cap);        // column lists gather the row objects' vectors
                s.kn_pw = ((p->types[s.col].n + s.kn_pc - 1) / s.kn_pc + 63) / 64 * 64;
                s.kn_ph = ((s.nr + s.kn_pr - 1) / s.kn_pr + 63) / 64 * 64;
                p->types[s.row].keep_prev = p->types[s.col].keep_prev = true;
                if (p->bf16) p->types[s.row].need_rows = true;              // the column lists gather the row type's bf16 rows
            }
            // sparse 0/1 relations as lists over bf16 factor rows (srp_bf16_v6_kernel<.., SRP_ONES>): both ranks 64 / 128 / 256
            auto gather_rank = [](int c) { return c == 64 || c == 128 || c == 256; };
            for (RelState& s : p->rels) {
                s.sp_gather = p->bf16 && s.binary && !s.masked && !s.absent && gather_rank(p->types[s.row].c) &&
                              gather_rank(p->types[s.col].c);
                if (!s.sp_gather) continue;
                p->types[s.row].need_rows = p->types[s.col].need_rows = true;
                // parts by the size of the gathered matrix alone (the number of ones is known at bind time, which may lower them)
                auto by_bytes = [&](int64_t n_in, int c) {
                    if (parts_env) return parts_env;
                    int q = 1;
                    while (q < 8 && (double)n_in * c * 2.0 / q > 3.5 * 1048576.0) q *= 2;
                    return q;
                };
                s.sp_pc = by_bytes(p->types[s.col].n, p->types[s.col].c);    // P: rows of G_j by the column index
                s.sp_pr = by_bytes(p->types[s.row].n, p->types[s.row].c);    // Q: rows of G_i by the row index
            }
        }
        p->thetas.resize(n_thetas);
        for (int t = 0; t < n_thetas; ++t) {
            if (thetas[t].type < 0 || thetas[t].type >= n_types || !thetas[t].data ||
                thetas[t].ld < p->types[thetas[t].type].n)
                SKF_FAIL(SKF_E_INVALID, "constraint %d invalid", t);
            if (p->variant == SKF_TRANSFORM && thetas[t].type != p->target)
                SKF_FAIL(SKF_E_INVALID, "constraint %d must be on the target object type", t);
            p->thetas[t].type = thetas[t].type;
            p->thetas[t].data = thetas[t].data;
            p->thetas[t].ld = thetas[t].ld;
            const int64_t nn = p->types[thetas[t].type].n;
            if (thetas[t].nnz < 0) SKF_FAIL(SKF_E_INVALID, "constraint %d: negative non-zero bound", t);
            if (thetas[t].nnz > 0 && thetas[t].nnz <= nn * nn / SKF_THETA_SPARSE_DIV) {
                p->thetas[t].sparse = true;
                p->thetas[t].nnz_cap = thetas[t].nnz;
            }
        }
        if (p->owned) {
            // SKF_BF16: the other owners' rows of a factor are read as bf16 operands only -- unless a constraint on the type
            // multiplies the f32 rows (theta_spmm_kernel / dense Theta).  Every type keeps its bf16 rows (the gathered form).
            for (TypeState& t : p->types) {
                t.gather_master = !p->bf16;
                if (p->bf16) t.need_rows = true;
            }
            for (const ThetaState& th : p->thetas) p->types[th.type].gather_master = true;
            // ... and the types of a masked relation (the same on every process, whatever it holds of the relation): the
            // known-entry form multiplies the f32 rows of both factors (cross-Gram matrices, T = G_j S^T, the dense part of Q)
            for (const RelState& r : p->rels)
                if (r.masked) p->types[r.row].gather_master = p->types[r.col].gather_master = true;
        }
        // ---- workspace layout: n-sized buffers in the master type, every c x c matrix in f64
        const size_t es = p->esz;
        size_t part_bytes = 0;
        size_t sp_part_bytes = 0;
        auto want_part = [&](int M, int N, int K, bool out_f64, bool sym = false) {
            TileCfg t = pick_tile(out_f64, p->engine, M, N);
            int sl = pick_splits(t, M, N, K);
            if (sym) sl = std::max(sl, pick_splits(t, M, N, K, true));     // (X^T X on the tiles on / below the diagonal: fewer tiles, more slices)
            if (!p->bf16 && (int64_t)cdiv(M, t.bm) * cdiv(N, t.bn) >= 256) {      // (relation contractions of the f32 / f64 engines)
                const int rs = pick_splits_relation(t, M, N, K, out_f64);
                if (rs > sl) sl = rs;
            }
            size_t need = (size_t)sl * (size_t)M * (size_t)N * (out_f64 ? 8 : 4);
            if (need > part_bytes) part_bytes = need;
        };
        int maxn = 2;
        // the E / D accumulators of all types form ONE contiguous range of the workspace, so that
        // a relation-sharded run sums them over the ranks with a single all-reduce
        // Three regions with the SAME internal layout -- all E, all D, all G (every type at the same offset in each) -- so
        // that the distributed iteration can cut them into `world` equal element ranges: reduce-scatter of E and of D,
        // update of the owned range of G, all-gather of G (skf_iterate_dist).  A pad behind each region takes the rounding
        // of the last range.
        p->acc_off = p->ws_bytes;
        for (int region = 0; region < 3; ++region) {
            const size_t begin = p->ws_bytes;
            for (int i = 0; i < n_types; ++i) {
                TypeState& t = p->types[i];
                const bool active = (p->variant != SKF_TRANSFORM) || i == p->target;
                if (region < 2 && !active) continue;
                add_slot(p, region == 0 ? t.E : region == 1 ? t.D : t.G, (size_t)t.n_alloc * t.c * es);
            }
            const size_t bytes = p->ws_bytes - begin;
            if (region == 0) { p->flat_e_off = begin; p->flat_bytes = bytes; }
            if (region == 1) p->flat_d_off = begin;
            if (region == 2) p->flat_g_off = begin;
            add_slot(p, p->flat_pad[region], 64 * 1024);
            if (region == 1) p->acc_bytes = p->ws_bytes - p->acc_off;
        }
        // the Gram matrices of all types form one range (SKF_OPT_OWNED_ROWS: one all-reduce of the partial sums)
        p->xg_off = p->ws_bytes;
        for (int i = 0; i < n_types; ++i) add_slot(p, p->types[i].Gram, (size_t)p->types[i].c * p->types[i].c * 8);
        p->xg_bytes = p->ws_bytes - p->xg_off;
        for (int i = 0; i < n_types; ++i) {
            TypeState& t = p->types[i];
            const bool active = (p->variant != SKF_TRANSFORM) || i == p->target;
            (void)active;
            if (p->bf16) {
                t.ldgt = pad64(t.n);
                add_slot(p, t.GTb, (size_t)t.c * t.ldgt * 2);
            }
            want_part(t.c, t.c, (int)t.n, true, true);
            if (t.keep_prev) add_slot(p, t.Gp, (size_t)t.n * t.c * es);
            if (p->bf16 && t.need_rows) {
                t.ldrow = (t.c + 7) / 8 * 8;
                add_slot(p, t.Grow, ((size_t)t.n_alloc + 1) * t.ldrow * 2);     // (+ 1: the all-zero row of the v6 list kernel)
            }
            if (p->variant != SKF_TRANSFORM) {
                add_slot(p, t.K, (size_t)t.c * t.c * 8);
                if (!p->f64) {
                    add_slot(p, t.Bp32, (size_t)t.c * t.c * 4);
                    add_slot(p, t.Bn32, (size_t)t.c * t.c * 4);
                }
                if (t.n_pad > maxn) maxn = t.n_pad;
            } else if (i == p->target) {
                add_slot(p, t.Ec, (size_t)t.n * t.c * es);
                add_slot(p, t.Dc, (size_t)t.n * t.c * es);
                add_slot(p, t.Galt, (size_t)t.n * t.c * es);
                add_slot(p, t.Bp_tot, (size_t)t.c * t.c * 8);
                add_slot(p, t.Bn_tot, (size_t)t.c * t.c * 8);
            }
        }
        size_t sq_elems = 1;
        if (p->variant != SKF_TRANSFORM) {
            // the per-type sums of B+ / B- form one range: one memset per iteration clears them all
            p->btot_off = p->ws_bytes;
            for (TypeState& t : p->types) {
                add_slot(p, t.Bp_tot, (size_t)t.c * t.c * 8);
                add_slot(p, t.Bn_tot, (size_t)t.c * t.c * 8);
            }
            p->btot_bytes = p->ws_bytes - p->btot_off;
            // exchange ranges of row-block sharding: all W; then Q of unmasked, then of masked relations
            p->xw_off = p->ws_bytes;
            for (RelState& r : p->rels) add_slot(p, r.W, (size_t)p->types[r.row].c * p->types[r.col].c * 8);
            p->xw_bytes = p->ws_bytes - p->xw_off;
            p->xq_off = p->ws_bytes;
            for (RelState& r : p->rels)
                if (!r.masked) add_slot(p, r.Q, (size_t)p->types[r.col].n_alloc * p->types[r.row].c * es);
            p->xq_bytes = p->ws_bytes - p->xq_off;
            p->xqm_off = p->ws_bytes;
            for (RelState& r : p->rels)
                if (r.masked) add_slot(p, r.Q, (size_t)p->types[r.col].n_alloc * p->types[r.row].c * es);
            p->xqm_bytes = p->ws_bytes - p->xqm_off;
        }
        for (RelState& r : p->rels) {
            TypeState& ti = p->types[r.row];
            TypeState& tj = p->types[r.col];
            const size_t cc = (size_t)ti.c * tj.c * 8;
            const int64_t nr = p->variant == SKF_TRANSFORM ? ti.n : r.nr;      // local rows
            if (p->variant == SKF_TRANSFORM) { r.nr = ti.n; r.r0 = 0; }
            add_slot(p, r.S, cc);
            add_slot(p, r.U, cc);
            if (nr > 0 && !r.kn) add_slot(p, r.H, (size_t)nr * tj.c * es);
            if (nr > 0 && !r.kn && (p->variant != SKF_TRANSFORM || r.row == p->target)) add_slot(p, r.P, (size_t)nr * tj.c * es);
            if (p->variant == SKF_TRANSFORM && r.col == p->target) add_slot(p, r.Q, (size_t)tj.n * ti.c * es);
            if (p->variant != SKF_TRANSFORM) {
                add_slot(p, r.T1, cc);
                if (!p->f64) add_slot(p, r.S32, cc / 2);
                if (nr > 0) want_part(ti.c, tj.c, (int)nr, true);
            }
            want_part(ti.c, tj.c, ti.c > tj.c ? ti.c : tj.c, true);
            want_part(ti.c, ti.c, tj.c, true);
            want_part(tj.c, tj.c, ti.c, true);
            if (nr <= 0 && r.kn) add_slot(p, r.U2, cc);             // (owned rows, none of this relation's here: the dense part of Q
                                                                    //  is still added on this process's rows of the column type)
            if (nr <= 0) continue;
            if (r.kn) {
                // the known entries as row lists and column lists, the gathered vectors, the row-side product, c x c scratch
                const size_t cap = (size_t)r.kn_cap;
                r.ldmb = (tj.n + 127) / 128 * 16;
                add_slot(p, r.Mb, (size_t)nr * r.ldmb);                         // (bind time only)
                add_slot(p, r.KrPtr, ((size_t)nr * r.kn_pc + 1) * 8);
                add_slot(p, r.KrIdx, cap * 4);
                add_slot(p, r.KrVal, cap * es);
                add_slot(p, r.KcPtr, ((size_t)tj.n * r.kn_pr + 1) * 8);
                add_slot(p, r.KcIdx, cap * 4);
                add_slot(p, r.KcVal, cap * es);
                add_slot(p, r.KcE, cap * es);
                const size_t cnt = std::max((size_t)nr * r.kn_pc, 2 * (size_t)tj.n * r.kn_pr);
                add_slot(p, r.KCnt, cnt * 4);
                r.kn_ldf = p->bf16 ? (ti.c + 7) / 8 * 8 : ti.c;
                if (p->bf16) add_slot(p, r.FiB, ((size_t)tj.n + 1) * r.kn_ldf * 2);  // (+ 1: the all-zero row of the v6 list kernel;
                                                                                    //  the row type's vectors: ti.Grow)
                add_slot(p, r.Tm, (size_t)tj.n * ti.c * es);
                add_slot(p, r.A, (size_t)nr * ti.c * es);
                if (r.kn_pc > 1) add_slot(p, r.Apart, (size_t)r.kn_pc * nr * ti.c * es);
                if (r.kn_pr > 1) add_slot(p, r.Qpart, (size_t)r.kn_pr * tj.n * ti.c * es);
                add_slot(p, r.Sp, cc);
                add_slot(p, r.U2, cc);
                add_slot(p, r.Xi, (size_t)ti.c * ti.c * 8);
                add_slot(p, r.Xj, (size_t)tj.c * tj.c * 8);
                add_slot(p, r.Bf, (size_t)ti.c * ti.c * 8);
                want_part(ti.c, tj.c, (int)tj.n, true);                         // W = Y^T G_j
                want_part(ti.c, ti.c, (int)nr, true);                           // G_i'^T G_i
                want_part(tj.c, tj.c, (int)tj.n, true);                         // G_j^T G_j'
                want_part((int)tj.n, ti.c, tj.c, p->f64);                       // T = G_j S^T ; Q += G_j (S^T Gram_i)
                want_part((int)nr, ti.c, ti.c, p->f64);                         // A += G_i (S Gram_j S^T)
                const size_t waves = (size_t)r.kn_pr * ((size_t)tj.n + 32) + 64;   // error partials: one per wave of the column pass
                if (waves > sq_elems) sq_elems = waves;
                continue;
            }
            if (r.mask && !p->bf16) add_slot(p, r.Rw, (size_t)nr * tj.n * es);
            if (r.mask) {
                r.ldmb = (tj.n + 127) / 128 * 16;               // bytes per packed mask row: whole 128-column tiles
                add_slot(p, r.Mb, (size_t)nr * r.ldmb);
            }
            if (p->bf16 && r.mask) {
                // known entries as compact per-tile lists, up to 1/8 of the relation (beyond that the completion
                // blends through the mask): 4 bytes per entry = at most a quarter of the bf16 relation's bytes
                const size_t tiles = (size_t)cdiv(nr, 256) * cdiv(tj.n, 128);     // (128-column tiles: the finer grid)
                r.kcap = (size_t)nr * tj.n / 8 + 4096;
                add_slot(p, r.Kcnt, tiles * 4);
                add_slot(p, r.Koff, (tiles + 1) * 4);
                add_slot(p, r.Klist, r.kcap * 4);
            }
            if (p->bf16) {                                       // completion / residual tiles
                r.ldhb = pad64(tj.c);
                add_slot(p, r.Hb, (size_t)nr * r.ldhb * 2);
                add_slot(p, r.Gb, (size_t)tj.n * r.ldhb * 2);
            }
            if (p->bf16) {
                r.ldrb = pad64(tj.n);
                r.kq = pad64(nr);
                if (r.binary) {
                    r.ldbb = r.ldrb / 8;
                    add_slot(p, r.Bb, (size_t)r.kq * r.ldbb);
                    if (!r.masked) {                       // room for the CSR / CSC form of a sparse relation
                        const int64_t per = r.sp_gather ? 80 : 256;            // (one entry in 80 / in 256 set, see build_sparse_pattern)
                        r.sp_cap = (int64_t)nr * tj.n / per > 0 ? (int64_t)nr * tj.n / per : 1;
                        if (r.sp_gather) {
                            if (r.sp_pc > 1) add_slot(p, r.SpRpP, ((size_t)nr * r.sp_pc + 1) * 8);
                            if (r.sp_pr > 1) add_slot(p, r.SpCpP, ((size_t)tj.n * r.sp_pr + 1) * 8);
                            const size_t pb = std::max(r.sp_pc > 1 ? (size_t)r.sp_pc * nr * tj.c * 4 : (size_t)0,
                                                       r.sp_pr > 1 ? (size_t)r.sp_pr * tj.n * ti.c * 4 : (size_t)0);
                            if (pb > sp_part_bytes) sp_part_bytes = pb;
                        }
                        add_slot(p, r.SpRp, (size_t)(nr + 1) * 8);
                        add_slot(p, r.SpCp, (size_t)(tj.n + 1) * 8);
                        add_slot(p, r.SpCi, (size_t)r.sp_cap * 4);
                        add_slot(p, r.SpRi, (size_t)r.sp_cap * 4);
                        add_slot(p, r.SpCnt, (size_t)(nr + 2 * tj.n + 8) * 4);
                    }
                } else {
                    add_slot(p, r.Rb, (size_t)r.kq * r.ldrb * 2);
                }
                size_t b1 = bf16_part_bytes((int)nr, tj.c, (int)r.ldrb, r.binary), b2 = bf16_part_bytes((int)tj.n, ti.c, (int)r.kq, true);
                if (b1 > part_bytes) part_bytes = b1;
                if (b2 > part_bytes) part_bytes = b2;
            }
            want_part((int)nr, tj.c, (int)tj.n, p->f64);
            want_part((int)tj.n, ti.c, (int)nr, p->f64);
            want_part((int)nr, ti.c, tj.c, p->f64);
            want_part((int)tj.n, tj.c, ti.c, p->f64);
            size_t blocks = (size_t)cdiv(nr, 32) * cdiv(tj.n, 32);           // smallest tile any engine uses
            if (blocks > sq_elems) sq_elems = blocks;
        }
        // small graphs: every rank <= 64, sparse constraints only, a few thousand objects per type -> the fused schedule
        {
            bool ok = p->variant == SKF_DFMF && p->engine == SKF_ENGINE_MFMA && !p->bf16 && !p->sliced && n_types <= SM_MAXT &&
                      n_relations >= 1 && n_relations <= SM_MAXR && n_thetas <= SM_MAXTH;
            // (object counts: the Q shares of the schedule grow with n_i / 256 * n_j * c_i, and from a few thousand objects
            // on the relation contractions are worth the big tiles of the general schedule)
            for (const TypeState& t : p->types) ok = ok && t.c <= SMALLC && t.n <= SM_MAX_OBJECTS;
            for (const ThetaState& th : p->thetas) ok = ok && th.sparse;
            for (const RelState& r : p->rels) ok = ok && !r.absent && !r.masked;
            p->small_fused = ok;
            if (ok) {
                size_t wdoubles = 0, gdoubles = 0;
                for (size_t k = 0; k < p->rels.size(); ++k) {
                    RelState& r = p->rels[k];
                    const TypeState& ti = p->types[r.row];
                    const TypeState& tj = p->types[r.col];
                    int part = 0;
                    for (int64_t r0 = 0; r0 < ti.n; r0 += 64) p->sm_j1.push_back(SmJob{SMJ_P, (int)k, (int)r0, (int)std::min<int64_t>(64, ti.n - r0), part++, 0, 0, 0});
                    // Q = R^T G_i: a long inner dimension (the rows of the relation) over a small output -> shares of SM_QROWS rows
                    int qpart = 0;
                    for (int64_t k0 = 0; k0 < ti.n; k0 += SM_QROWS, ++qpart)
                        for (int64_t c0 = 0; c0 < tj.n; c0 += 64)
                            p->sm_j1.push_back(SmJob{SMJ_Q, (int)k, (int)c0, (int)std::min<int64_t>(64, tj.n - c0), qpart, (int)k0,
                                                     (int)std::min<int64_t>(SM_QROWS, ti.n - k0), 0});
                    r.sm_qparts = qpart;
                    wdoubles += align_up((size_t)part * ti.c * tj.c, 16);      // (shares of different relations / types never share a cache line)
                    add_slot(p, r.SmQ, (size_t)qpart * tj.n * ti.c * es);
                    add_slot(p, r.SmBp, (size_t)ti.c * ti.c * 8);
                    add_slot(p, r.SmBn, (size_t)ti.c * ti.c * 8);
                    add_slot(p, r.SmDp, (size_t)tj.c * tj.c * 8);
                    add_slot(p, r.SmDn, (size_t)tj.c * tj.c * 8);
                }
                for (size_t i = 0; i < p->types.size(); ++i) {
                    const TypeState& t = p->types[i];
                    int part = 0;
                    for (int64_t r0 = 0; r0 < t.n; r0 += SM_GROWS) p->sm_j1.push_back(SmJob{SMJ_GRAM, (int)i, (int)r0, (int)std::min<int64_t>(SM_GROWS, t.n - r0), part++, 0, 0, 0});
                    gdoubles += align_up((size_t)part * t.c * t.c, 16);
                    for (int64_t r0 = 0; r0 < t.n; r0 += 64) p->sm_j3.push_back(SmJob{0, (int)i, (int)r0, (int)std::min<int64_t>(64, t.n - r0), 0, 0, 0, 0});
                    bool constrained = false;
                    for (const ThetaState& th : p->thetas) constrained = constrained || th.type == (int)i;
                    if (constrained)      // one wave per row, four rows per workgroup
                        for (int64_t r0 = 0; r0 < t.n; r0 += 4) p->sm_j1.push_back(SmJob{SMJ_THETA, (int)i, (int)r0, (int)std::min<int64_t>(4, t.n - r0), 0, 0, 0, 0});
                }
                // Gram shares first (the inverse hangs off the last of them), then P (W hangs off the last of those), Q, constraints
                auto prio = [](const SmJob& j) { return j.kind == SMJ_GRAM ? 0 : (j.kind == SMJ_P ? 1 : (j.kind == SMJ_Q ? 2 : 3)); };
                std::stable_sort(p->sm_j1.begin(), p->sm_j1.end(), [&](const SmJob& a, const SmJob& b) { return prio(a) < prio(b); });
                add_slot(p, p->sm_tables, sizeof(SmTables));
                add_slot(p, p->sm_jobs1, p->sm_j1.size() * sizeof(SmJob));
                add_slot(p, p->sm_jobs3, p->sm_j3.size() * sizeof(SmJob));
                add_slot(p, p->sm_wpart, wdoubles * 8);
                add_slot(p, p->sm_gpart, gdoubles * 8);
                add_slot(p, p->sm_tickets, (p->types.size() + p->rels.size()) * sizeof(int));
                add_slot(p, p->sm_batch, SKF_MAX_BATCH * sizeof(void*));          // table of tables: [0] = this plan's
            }
        }
        size_t theta_tmp_bytes = 0;
        for (ThetaState& th : p->thetas) {
            TypeState& t = p->types[th.type];
            if (th.sparse) {
                add_slot(p, th.Cnt, (size_t)t.n * sizeof(int));
                add_slot(p, th.Rp, (size_t)(t.n + 1) * sizeof(int64_t));
                add_slot(p, th.Ci, (size_t)th.nnz_cap * sizeof(int));
                add_slot(p, th.Vv, (size_t)th.nnz_cap * es);
                continue;
            }
            want_part((int)t.n, t.c, (int)t.n, p->f64);
            if (p->bf16) {
                th.ldb = pad64(t.n);
                add_slot(p, th.Pb, (size_t)t.n * th.ldb * 2);
                add_slot(p, th.Nb, (size_t)t.n * th.ldb * 2);
                const size_t b = bf16_part_bytes((int)t.n, t.c, (int)th.ldb, false);
                if (b > part_bytes) part_bytes = b;
                if ((size_t)t.n * t.c * 4 > theta_tmp_bytes) theta_tmp_bytes = (size_t)t.n * t.c * 4;
            }
        }
        if (!p->thetas.empty()) add_slot(p, p->theta_flags, p->thetas.size() * 2 * sizeof(int));
        if (theta_tmp_bytes) add_slot(p, p->theta_tmp, theta_tmp_bytes);
        {   // the Gram products of all types side by side in one launch (gram_all): every product's slices at once
            std::vector<std::pair<int, int64_t>> cn;
            for (const TypeState& t : p->types) cn.emplace_back(t.c, t.n);
            if (cn.size() >= 2 && cn.size() <= 4) part_bytes = std::max(part_bytes, gram_group_bytes(p->engine, cn));
        }
        p->part_bytes = part_bytes;
        add_slot(p, p->part, part_bytes);
        if (sp_part_bytes) add_slot(p, p->sp_part, sp_part_bytes);
        size_t aux_bytes = 0;
        for (TypeState& t : p->types) {
            TileCfg tc = pick_tile(true, p->engine, t.c, t.c);
            const int sl = std::max(pick_splits(tc, t.c, t.c, (int)t.n), pick_splits(tc, t.c, t.c, (int)t.n, true));   // (gram(.., on_aux): symmetric)
            size_t need = (size_t)sl * (size_t)t.c * t.c * 8;
            if (need > aux_bytes) aux_bytes = need;
        }
        for (RelState& r : p->rels) {          // W = G_i^T P on the second stream (pipelined schedule)
            const int ci = p->types[r.row].c, cj = p->types[r.col].c;
            TileCfg tc = pick_tile(true, p->engine, ci, cj);
            size_t need = (size_t)pick_splits(tc, ci, cj, (int)r.nr) * (size_t)ci * cj * 8;
            if (need > aux_bytes) aux_bytes = need;
        }
        {
            std::vector<std::pair<int, int64_t>> cn;
            for (const TypeState& t : p->types) cn.emplace_back(t.c, t.n);
            if (cn.size() >= 2 && cn.size() <= 4) aux_bytes = std::max(aux_bytes, gram_group_bytes(p->engine, cn));
        }
        p->part_aux_bytes = aux_bytes;
        add_slot(p, p->part_aux, aux_bytes);
        p->sq_elems = sq_elems;
        add_slot(p, p->sqpart, sq_elems * 8);
        if (p->variant != SKF_TRANSFORM) {
            p->eig_maxn = maxn;
            p->eig_stride = (int64_t)maxn * maxn;
            const size_t mat = (size_t)p->eig_stride * n_types * sizeof(double);
            add_slot(p, p->eigA, mat);
            add_slot(p, p->eigV, mat);
            add_slot(p, p->eigVs, mat);
            add_slot(p, p->eigW, (size_t)maxn * n_types * sizeof(double));
            add_slot(p, p->eigN, (size_t)n_types * sizeof(int));
            add_slot(p, p->eigNorig, (size_t)n_types * sizeof(int));
            add_slot(p, p->eigOk, (size_t)n_types * sizeof(int));
            if (maxn > SWEEP_MAXN && n_types <= PINV_MAXB) add_slot(p, p->eigX, defl_scratch_bytes(n_types, p->eig_stride));
        }
        guard.p = nullptr;
        *out = p;
    });
}

int skf_plan_destroy(skf_plan* plan) {
    delete plan;
    return SKF_OK;
}

int skf_plan_workspace_bytes(const skf_plan* plan, size_t* bytes) {
    return guarded([&] {
        if (!plan || !bytes) SKF_FAIL(SKF_E_INVALID, "null argument");
        *bytes = plan->ws_bytes;
    });
}

// CSR + CSC of a very sparse binary relation from its bitmap (bind time): per-row counts on the device, prefix sums on the
// host; kept only when the ones fit the slots sized at plan creation (1 entry in 256)
static void build_sparse_pattern(skf_plan* p, RelState& r, hipStream_t st) {
    r.sparse = false;
    if (r.sp_cap <= 0 || !r.SpRp.ptr) return;
    const int64_t rows = r.nr, cols = p->types[r.col].n;
    if (rows <= 0 || cols <= 0) return;
    const int wgrid = (int)((rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048);
    int* rowcnt = (int*)r.SpCnt.ptr;
    int* colcnt = rowcnt + rows;
    int* fillpos = colcnt + cols;
    hipLaunchKernelGGL(bits_row_count_kernel, dim3(wgrid), dim3(256), 0, st, (const uint8_t*)r.Bb.ptr, r.ldbb, rows, rowcnt);
    check_launch("bits_row_count");
    std::vector<int> cnt((size_t)(rows > cols ? rows : cols));
    SKF_HIP(hipMemcpyAsync(cnt.data(), rowcnt, (size_t)rows * 4, hipMemcpyDeviceToHost, st));
    SKF_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> ptr((size_t)(rows > cols ? rows : cols) + 1);
    int64_t tot = 0;
    for (int64_t k = 0; k < rows; ++k) { ptr[k] = tot; tot += cnt[k]; }
    ptr[rows] = tot;
    // the form by the count: lists over bf16 factor rows (srp_bf16_v6_kernel<.., SRP_ONES>) up to 1 entry in 80 -- measured
    // at config 5 (profiles/r03_srp_v6.txt): ~14-22 ps per one and contraction against ~0.28 ps per CELL of the bitmap
    // kernels, break-even near 1 in 64 --; without that form (other ranks) lists over the f32 rows up to 1 in 256
    if (tot > r.sp_cap || (!r.sp_gather && tot > rows * cols / 256)) return;
    r.sp_nnz = tot;
    SKF_HIP(hipMemcpyAsync(r.SpRp.ptr, ptr.data(), (size_t)(rows + 1) * 8, hipMemcpyHostToDevice, st));
    SKF_HIP(hipMemsetAsync(colcnt, 0, (size_t)cols * 2 * 4, st));
    SKF_HIP(hipStreamSynchronize(st));                       // (`ptr` is reused below)
    if (tot > 0) {
        hipLaunchKernelGGL(bits_csr_fill_kernel, dim3(wgrid), dim3(256), 0, st, (const uint8_t*)r.Bb.ptr, r.ldbb, rows,
                           (const int64_t*)r.SpRp.ptr, (int*)r.SpCi.ptr);
        hipLaunchKernelGGL(csr_col_count_kernel, dim3(elem_grid(tot)), dim3(256), 0, st, (const int*)r.SpCi.ptr, tot, colcnt);
        check_launch("bits_csr_fill");
    }
    SKF_HIP(hipMemcpyAsync(cnt.data(), colcnt, (size_t)cols * 4, hipMemcpyDeviceToHost, st));
    SKF_HIP(hipStreamSynchronize(st));
    int64_t t2 = 0;
    for (int64_t k = 0; k < cols; ++k) { ptr[k] = t2; t2 += cnt[k]; }
    ptr[cols] = t2;
    SKF_HIP(hipMemcpyAsync(r.SpCp.ptr, ptr.data(), (size_t)(cols + 1) * 8, hipMemcpyHostToDevice, st));
    if (tot > 0) {
        hipLaunchKernelGGL(csr_transpose_fill_kernel, dim3(wgrid), dim3(256), 0, st, (const int64_t*)r.SpRp.ptr,
                           (const int*)r.SpCi.ptr, rows, (const int64_t*)r.SpCp.ptr, fillpos, (int*)r.SpRi.ptr);
        hipLaunchKernelGGL(csc_sort_kernel, dim3(elem_grid(cols)), dim3(256), 0, st, (const int64_t*)r.SpCp.ptr,
                           (int*)r.SpRi.ptr, cols);
        check_launch("csc_build");
    }
    if (r.sp_gather) {          // lists in parts pinned to XCDs, as long as a segment still holds a batch of entries
        const bool forced = p->sw.known_parts_forced;                    // (tests: short lists in parts too)
        auto fit = [&](int q, int64_t n_out) {
            while (!forced && q > 1 && (double)tot / ((double)n_out * q) < 64.0) q /= 2;
            return q;
        };
        r.sp_pc = fit(r.SpRpP.ptr ? r.sp_pc : 1, rows);
        r.sp_pr = fit(r.SpCpP.ptr ? r.sp_pr : 1, cols);
        r.sp_pw = ((cols + r.sp_pc - 1) / r.sp_pc + 63) / 64 * 64;
        r.sp_ph = ((rows + r.sp_pr - 1) / r.sp_pr + 63) / 64 * 64;
        if (r.sp_pc > 1)
            hipLaunchKernelGGL(parted_ptr_kernel, dim3(elem_grid(rows * r.sp_pc + 1)), dim3(256), 0, st, (const int64_t*)r.SpRp.ptr,
                               (const int*)r.SpCi.ptr, rows, r.sp_pc, r.sp_pw, (int64_t*)r.SpRpP.ptr);
        if (r.sp_pr > 1)
            hipLaunchKernelGGL(parted_ptr_kernel, dim3(elem_grid(cols * r.sp_pr + 1)), dim3(256), 0, st, (const int64_t*)r.SpCp.ptr,
                               (const int*)r.SpRi.ptr, cols, r.sp_pr, r.sp_ph, (int64_t*)r.SpCpP.ptr);
        check_launch("parted_ptr");
    }
    SKF_HIP(hipStreamSynchronize(st));
    r.sparse = true;
}

int skf_plan_bind_workspace(skf_plan* p, void* ws, size_t bytes, void* stream) {
    return guarded([&] {
        if (!p || !ws) SKF_FAIL(SKF_E_INVALID, "null argument");
        if (bytes < p->ws_bytes) SKF_FAIL(SKF_E_WORKSPACE, "workspace %zu B < required %zu B", bytes, p->ws_bytes);
        if (((uintptr_t)ws & 255) != 0) SKF_FAIL(SKF_E_WORKSPACE, "workspace must be 256-byte aligned");
        for (Slot* s : p->slots) s->ptr = (char*)ws + s->off;
        p->ws_base = ws;
        p->sw = Switches::read();          // the only place a plan looks at the environment
        hipStream_t st = as_stream(stream);
        for (RelState& r : p->rels) {
            if (!r.mask) continue;
            // the mask in the engine's layout: one bit per entry, rows padded to whole 128-column tiles.
            // The caller's mask (bytes or bits) is not referenced after this call.
            const int64_t rows = r.nr, cols = p->types[r.col].n;
            if (r.mask_is_bits) {
                SKF_HIP(hipMemsetAsync(r.Mb.ptr, 0, r.Mb.bytes, st));
                hipLaunchKernelGGL(copy_mask_bits_kernel, dim3(elem_grid(rows * ((cols + 7) / 8))), dim3(256), 0, st,
                                   (uint8_t*)r.Mb.ptr, r.ldmb, r.mask, r.ldmask, rows, cols);
            } else {
                hipLaunchKernelGGL(pack_mask_kernel, dim3(elem_grid(rows * r.ldmb)), dim3(256), 0, st, (uint8_t*)r.Mb.ptr,
                                   r.ldmb, r.mask, r.ldmask, rows, cols);
            }
            check_launch("pack_mask");
            if (r.kn) {                            // the known entries as lists; no working copy of the relation
                build_known_lists(p, r, st);
                continue;
            }
            if (p->bf16) {
                // known entries of every tile of the completion pass (256 rows x epi_tile columns) as a compact list
                // (count, prefix sum on the host, fill)
                const int tx = cdiv(rows, 256), ty = cdiv(cols, 128);
                const size_t tiles = (size_t)tx * ty;
                KnownArgs ka;
                ka.mbits = (const uint8_t*)r.Mb.ptr; ka.ldmb = r.ldmb;
                ka.Rin = (const uint16_t*)r.R_in; ka.ldin = r.ld_in;
                ka.rows = (int)rows; ka.cols = (int)cols;
                ka.tile_cols = 128;
                ka.counts = (uint32_t*)r.Kcnt.ptr; ka.off = nullptr; ka.list = nullptr;
                hipLaunchKernelGGL(known_entries_kernel, dim3(tx, ty), dim3(256), 0, st, ka);
                check_launch("known_entries(count)");
                std::vector<uint32_t> cnt(tiles), off(tiles + 1);
                SKF_HIP(hipMemcpyAsync(cnt.data(), r.Kcnt.ptr, tiles * 4, hipMemcpyDeviceToHost, st));
                SKF_HIP(hipStreamSynchronize(st));
                uint64_t tot = 0;
                for (size_t t = 0; t < tiles; ++t) { off[t] = (uint32_t)tot; tot += cnt[t]; }
                off[tiles] = (uint32_t)tot;
                r.use_klist = tot <= r.kcap && tot < 0xFFFFFFFFull;
                if (r.use_klist) {
                    SKF_HIP(hipMemcpyAsync(r.Koff.ptr, off.data(), (tiles + 1) * 4, hipMemcpyHostToDevice, st));
                    ka.off = (const uint32_t*)r.Koff.ptr; ka.list = (uint32_t*)r.Klist.ptr;
                    hipLaunchKernelGGL(known_entries_kernel, dim3(tx, ty), dim3(256), 0, st, ka);
                    check_launch("known_entries(fill)");
                    SKF_HIP(hipStreamSynchronize(st));          // `off` dies here; bind is not on the hot path
                }
                continue;                          // bf16: the padded copy below is the working set
            }
            copy2d(r.Rw.ptr, cols, r.R_in, r.ld_in, rows, cols, p->esz, st);
            r.R = r.Rw.ptr;
            r.ldr = cols;
        }
        if (p->bf16) {
            // the caller's bf16 relation is copied ONCE into a zero-padded row-major layout (rows to a multiple of
            // 64: the inner dimension of Q = R^T G_i; columns to a multiple of 64: the inner dimension of
            // P = R G_j); it is not referenced after this call
            for (TypeState& t : p->types) {
                SKF_HIP(hipMemsetAsync(t.GTb.ptr, 0, t.GTb.bytes, st));
                if (t.Grow.bytes) SKF_HIP(hipMemsetAsync(t.Grow.ptr, 0, t.Grow.bytes, st));
            }
            for (RelState& r : p->rels) {
                if (r.absent || r.kn) continue;
                const int64_t rows = r.nr, cols = p->types[r.col].n;
                if (r.binary) {
                    int* bad = (int*)p->sqpart.ptr;                  // (scratch word; bind is not on the hot path)
                    SKF_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
                    hipLaunchKernelGGL(pack_binary_kernel, dim3(elem_grid(r.kq * r.ldbb)), dim3(256), 0, st, (uint8_t*)r.Bb.ptr,
                                       r.ldbb, r.kq, (const uint16_t*)r.R_in, r.ld_in, rows, cols, bad);
                    check_launch("pack_binary");
                    int hbad = 0;
                    SKF_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
                    SKF_HIP(hipStreamSynchronize(st));
                    if (hbad) SKF_FAIL(SKF_E_INVALID, "a relation flagged SKF_REL_BINARY holds an entry that is neither 0 nor 1");
                    if (r.Hb.bytes) {
                        SKF_HIP(hipMemsetAsync(r.Hb.ptr, 0, r.Hb.bytes, st));
                        SKF_HIP(hipMemsetAsync(r.Gb.ptr, 0, r.Gb.bytes, st));
                    }
                    r.R = r.Bb.ptr;
                    r.ldr = r.ldrb;
                    build_sparse_pattern(p, r, st);
                    continue;
                }
                SKF_HIP(hipMemsetAsync(r.Rb.ptr, 0, r.Rb.bytes, st));
                if (r.Hb.bytes) {
                    SKF_HIP(hipMemsetAsync(r.Hb.ptr, 0, r.Hb.bytes, st));
                    SKF_HIP(hipMemsetAsync(r.Gb.ptr, 0, r.Gb.bytes, st));
                }
                launch_to_bf16<uint16_t>((uint16_t*)r.Rb.ptr, r.ldrb, (const uint16_t*)r.R_in, r.ld_in, rows, cols, false, st);
                r.R = r.Rb.ptr;
                r.ldr = r.ldrb;
            }
        }
        for (ThetaState& th : p->thetas) {
            if (!th.sparse) continue;
            // CSR of a sparse constraint: per-row counts on the device, prefix sum on the host, fill on the device
            const int64_t n = p->types[th.type].n;
            const int grid = (int)((n + 3) / 4 < 2048 ? (n + 3) / 4 : 2048);
            if (p->f64)
                hipLaunchKernelGGL((theta_row_count_kernel<double>), dim3(grid), dim3(256), 0, st, (const double*)th.data, th.ld, n, (int*)th.Cnt.ptr);
            else
                hipLaunchKernelGGL((theta_row_count_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)th.data, th.ld, n, (int*)th.Cnt.ptr);
            check_launch("theta_row_count");
            std::vector<int> cnt((size_t)n);
            SKF_HIP(hipMemcpyAsync(cnt.data(), th.Cnt.ptr, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st));
            SKF_HIP(hipStreamSynchronize(st));
            std::vector<int64_t> rp((size_t)n + 1);
            int64_t tot = 0;
            for (int64_t r = 0; r < n; ++r) { rp[(size_t)r] = tot; tot += cnt[(size_t)r]; }
            rp[(size_t)n] = tot;
            if (tot > th.nnz_cap)
                SKF_FAIL(SKF_E_INVALID, "constraint on type %d holds %lld non-zeros, more than the bound %lld given in skf_theta_desc.nnz",
                         th.type, (long long)tot, (long long)th.nnz_cap);
            th.nnz = tot;
            SKF_HIP(hipMemcpyAsync(th.Rp.ptr, rp.data(), ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
            if (p->f64)
                hipLaunchKernelGGL((theta_csr_fill_kernel<double>), dim3(grid), dim3(256), 0, st, (const double*)th.data, th.ld, n,
                                   (const int64_t*)th.Rp.ptr, (int*)th.Ci.ptr, (double*)th.Vv.ptr);
            else
                hipLaunchKernelGGL((theta_csr_fill_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)th.data, th.ld, n,
                                   (const int64_t*)th.Rp.ptr, (int*)th.Ci.ptr, (float*)th.Vv.ptr);
            check_launch("theta_csr_fill");
            SKF_HIP(hipStreamSynchronize(st));          // `rp` dies here; bind is not on the hot path
        }
        if (!p->thetas.empty()) {
            // which halves of every constraint's +- split are non-empty (one device pass, read back here:
            // bind is not on the hot path), and the bf16 engine's copies of the non-empty halves
            SKF_HIP(hipMemsetAsync(p->theta_flags.ptr, 0, p->theta_flags.bytes, st));
            for (size_t k = 0; k < p->thetas.size(); ++k) {
                ThetaState& th = p->thetas[k];
                if (th.sparse) continue;
                const int64_t n = p->types[th.type].n;
                int* fl = (int*)p->theta_flags.ptr + 2 * k;
                if (p->f64)
                    hipLaunchKernelGGL((sign_flags_kernel<double>), dim3(elem_grid(n * n)), dim3(256), 0, st,
                                       (const double*)th.data, th.ld, n, n, fl);
                else
                    hipLaunchKernelGGL((sign_flags_kernel<float>), dim3(elem_grid(n * n)), dim3(256), 0, st,
                                       (const float*)th.data, th.ld, n, n, fl);
                check_launch("sign_flags");
            }
            std::vector<int> flags(p->thetas.size() * 2);
            SKF_HIP(hipMemcpyAsync(flags.data(), p->theta_flags.ptr, flags.size() * sizeof(int), hipMemcpyDeviceToHost, st));
            SKF_HIP(hipStreamSynchronize(st));
            for (size_t k = 0; k < p->thetas.size(); ++k) {
                ThetaState& th = p->thetas[k];
                if (th.sparse) continue;
                th.has_pos = flags[2 * k] != 0;
                th.has_neg = flags[2 * k + 1] != 0;
                if (!p->bf16) continue;
                const int64_t n = p->types[th.type].n;
                for (int half = 0; half < 2; ++half) {
                    if (!(half == 0 ? th.has_pos : th.has_neg)) continue;
                    Slot& dst = half == 0 ? th.Pb : th.Nb;
                    SKF_HIP(hipMemsetAsync(dst.ptr, 0, dst.bytes, st));
                    hipLaunchKernelGGL(split_to_bf16_kernel, dim3(elem_grid(n * n)), dim3(256), 0, st, (uint16_t*)dst.ptr,
                                       th.ldb, (const float*)th.data, th.ld, n, n, half == 0 ? AOP_POS : AOP_NEG);
                    check_launch("split_to_bf16");
                }
            }
        }
        if (p->variant != SKF_TRANSFORM) {
            std::vector<int> n_pad, n_orig;
            for (TypeState& t : p->types) {
                n_pad.push_back(t.n_pad);
                n_orig.push_back(t.c);
            }
            SKF_HIP(hipMemcpyAsync(p->eigN.ptr, n_pad.data(), n_pad.size() * sizeof(int), hipMemcpyHostToDevice, st));
            SKF_HIP(hipMemcpyAsync(p->eigNorig.ptr, n_orig.data(), n_orig.size() * sizeof(int), hipMemcpyHostToDevice, st));
            SKF_HIP(hipStreamSynchronize(st));     // the host vectors die here; bind is not on the hot path
        }
        if (p->small_fused && (p->sw.no_small_fused || p->sw.no_small_chain)) p->small_fused = false;
        if (p->small_fused) {
            // job tables and the pointer table of the fused small-graph schedule (skf_small.h)
            SmTables tb;
            memset(&tb, 0, sizeof tb);
            tb.n_types = (int)p->types.size(); tb.n_rels = (int)p->rels.size(); tb.n_thetas = (int)p->thetas.size();
            tb.nan_upd = 1;                                   // DFMF: nan_to_num on the A / B / C / D terms (_dfmf.py:254-276)
            tb.wpart = (double*)p->sm_wpart.ptr; tb.gpart = (double*)p->sm_gpart.ptr;
            tb.tickets = (int*)p->sm_tickets.ptr;
            SKF_HIP(hipMemsetAsync(p->sm_tickets.ptr, 0, p->sm_tickets.bytes, st));
            tb.eigA = (double*)p->eigA.ptr; tb.eigV = (double*)p->eigV.ptr; tb.eigOk = (int*)p->eigOk.ptr;
            tb.eig_stride = p->eig_stride;
            tb.chol_thr = chol_rel_threshold(p->sw);
            tb.eig.A = (double*)p->eigA.ptr; tb.eig.V = (double*)p->eigV.ptr; tb.eig.Vs = (double*)p->eigVs.ptr;
            tb.eig.w = (double*)p->eigW.ptr; tb.eig.stride = p->eig_stride; tb.eig.wstride = p->eig_maxn;
            tb.eig.n = (const int*)p->eigN.ptr; tb.eig.n_orig = (const int*)p->eigNorig.ptr;
            tb.eig.chol_ok = (int*)p->eigOk.ptr;
            tb.eig.max_sweeps = 30;
            tb.defl_lo = deflation_lo(p->sw); tb.defl_hi = 1e-7;
            tb.lds_rank = p->eig_maxn < 64 ? p->eig_maxn : 64;          // packed r (r + 1) / 2 doubles inside the staging tiles
            tb.sweep_single = p->sw.small_sweep1 ? 1 : 0;
            int64_t goff = 0, woff = 0;
            for (size_t i = 0; i < p->types.size(); ++i) {
                TypeState& t = p->types[i];
                SmType& d = tb.t[i];
                d.G = t.G.ptr; d.E = t.E.ptr; d.D = t.D.ptr; d.Gram = (double*)t.Gram.ptr; d.K = (double*)t.K.ptr;
                d.n = t.n; d.c = t.c; d.gpart_off = goff; d.n_gjobs = (int)((t.n + SM_GROWS - 1) / SM_GROWS);
                goff += (int64_t)align_up((size_t)d.n_gjobs * t.c * t.c, 16);
                for (const ThetaState& th : p->thetas) d.has_theta = d.has_theta || th.type == (int)i;
            }
            for (size_t k = 0; k < p->rels.size(); ++k) {
                RelState& r = p->rels[k];
                SmRel& d = tb.r[k];
                d.R = r.R; d.ldr = r.ldr; d.P = r.P.ptr; d.Q = r.SmQ.ptr; d.n_qparts = r.sm_qparts; d.W = (double*)r.W.ptr; d.S = (double*)r.S.ptr;
                d.Bp = (double*)r.SmBp.ptr; d.Bn = (double*)r.SmBn.ptr; d.Dp = (double*)r.SmDp.ptr; d.Dn = (double*)r.SmDn.ptr;
                d.row = r.row; d.col = r.col; d.wpart_off = woff; d.n_pjobs = (int)((p->types[r.row].n + 63) / 64);
                woff += (int64_t)align_up((size_t)d.n_pjobs * p->types[r.row].c * p->types[r.col].c, 16);
            }
            for (size_t k = 0; k < p->thetas.size(); ++k) {
                ThetaState& th = p->thetas[k];
                tb.th[k].rp = (const int64_t*)th.Rp.ptr; tb.th[k].ci = (const int*)th.Ci.ptr; tb.th[k].vv = th.Vv.ptr;
                tb.th[k].type = th.type;
            }
            SKF_HIP(hipMemcpyAsync(p->sm_tables.ptr, &tb, sizeof tb, hipMemcpyHostToDevice, st));
            p->sm_batch_host.assign(1, p->sm_tables.ptr);
            SKF_HIP(hipMemcpyAsync(p->sm_batch.ptr, p->sm_batch_host.data(), sizeof(void*), hipMemcpyHostToDevice, st));
            SKF_HIP(hipMemcpyAsync(p->sm_jobs1.ptr, p->sm_j1.data(), p->sm_j1.size() * sizeof(SmJob), hipMemcpyHostToDevice, st));
            SKF_HIP(hipMemcpyAsync(p->sm_jobs3.ptr, p->sm_j3.data(), p->sm_j3.size() * sizeof(SmJob), hipMemcpyHostToDevice, st));
            SKF_HIP(hipStreamSynchronize(st));     // (`tb` dies here; bind is not on the hot path)
        }
        if (p->owned) {
            // padded layouts of the exchanges: rows past the objects of a type stay zero for good (they are gathered and
            // scattered with the rest), the Gram range is summed as a whole
            for (TypeState& t : p->types) {
                SKF_HIP(hipMemsetAsync(t.G.ptr, 0, t.G.bytes, st));
                SKF_HIP(hipMemsetAsync(t.E.ptr, 0, t.E.bytes, st));
                SKF_HIP(hipMemsetAsync(t.D.ptr, 0, t.D.bytes, st));
            }
            for (RelState& r : p->rels) SKF_HIP(hipMemsetAsync(r.Q.ptr, 0, r.Q.bytes, st));
            SKF_HIP(hipMemsetAsync((char*)ws + p->xg_off, 0, p->xg_bytes, st));
            SKF_HIP(hipMemsetAsync((char*)ws + p->xw_off, 0, p->xw_bytes, st));
            if (!p->cs && !p->sw.no_overlap && p->sw.comm_stream)
                SKF_HIP(hipStreamCreateWithFlags(&p->cs, hipStreamNonBlocking));
        }
        p->pipeline = !p->sw.no_pipeline;
        // (a plan on the three-launch schedule of small graphs issues everything on the caller's stream: no second stream to
        // create and destroy -- at ten restarts of the README graph the streams of the plans were 4 of 30 ms)
        if (p->variant != SKF_TRANSFORM && !p->aux && !p->small_fused) {
            if (!p->sw.no_overlap) {
                {   // the second stream at the LOWEST priority: its launches fill what the contractions of the main stream
                    // leave free instead of taking CUs from them (config 5 +0.9 %, config 3 +0.5 % against the default priority)
                    int lo = 0, hi = 0;
                    SKF_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
                    // (plans with owned rows, where the second stream carries the critical path of a rank: lowest / default /
                    // highest priority measured equal -- 2.58 / 2.56 / 2.55 ms for rank 3 of 8 at config 3 --, a running
                    // contraction workgroup is not preempted; profiles/r04_owned_rank_emulation.txt)
                    SKF_HIP(hipStreamCreateWithPriority(&p->aux, hipStreamNonBlocking, lo));
                }
                SKF_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
                SKF_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
                p->overlap = true;
            }
        }
        if (p->graph_exec) {
            (void)hipGraphExecDestroy(p->graph_exec);
            p->graph_exec = nullptr;
        }
        p->graph_failed = false;
        p->bound = true;
        p->prepared = false;
        p->first_iter = true;
        p->kn_first = true;
    });
}

static void check_bound(const skf_plan* p) {
    if (!p) SKF_FAIL(SKF_E_INVALID, "null plan");
    if (!p->bound) SKF_FAIL(SKF_E_STATE, "workspace not bound");
}

int skf_set_factor(skf_plan* p, int32_t type, const void* G, int64_t ld, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (type < 0 || type >= (int)p->types.size() || !G) SKF_FAIL(SKF_E_INVALID, "bad type index / pointer");
        TypeState& t = p->types[type];
        if (ld < t.c) SKF_FAIL(SKF_E_INVALID, "ld %lld < rank %d", (long long)ld, t.c);
        copy2d(t.G.ptr, t.c, G, ld, t.n, t.c, p->esz, as_stream(stream));
        refresh_gt(p, t, as_stream(stream));
        t.set = true;
        p->prepared = false;
    });
}

int skf_get_factor(const skf_plan* p, int32_t type, void* G, int64_t ld, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (type < 0 || type >= (int)p->types.size() || !G) SKF_FAIL(SKF_E_INVALID, "bad type index / pointer");
        const TypeState& t = p->types[type];
        if (ld < t.c) SKF_FAIL(SKF_E_INVALID, "ld %lld < rank %d", (long long)ld, t.c);
        copy2d(G, ld, t.G.ptr, t.c, t.n, t.c, p->esz, as_stream(stream));
    });
}

int skf_set_backbone(skf_plan* p, int32_t rel, const void* S, int64_t ld, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (rel < 0 || rel >= (int)p->rels.size() || !S) SKF_FAIL(SKF_E_INVALID, "bad relation index / pointer");
        RelState& r = p->rels[rel];
        const int ci = p->types[r.row].c, cj = p->types[r.col].c;
        if (ld < cj) SKF_FAIL(SKF_E_INVALID, "ld too small");
        if (p->f64) {
            copy2d(r.S.ptr, cj, S, ld, ci, cj, 8, as_stream(stream));
        } else {
            hipLaunchKernelGGL((cast_kernel<double, float>), dim3(elem_grid((int64_t)ci * cj)), dim3(256), 0,
                               as_stream(stream), (double*)r.S.ptr, (int64_t)cj, (const float*)S, ld, (int64_t)ci,
                               (int64_t)cj);
            check_launch("cast");
        }
        r.s_set = true;
        p->prepared = false;
    });
}

int skf_get_backbone(const skf_plan* p, int32_t rel, void* S, int64_t ld, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (rel < 0 || rel >= (int)p->rels.size() || !S) SKF_FAIL(SKF_E_INVALID, "bad relation index / pointer");
        const RelState& r = p->rels[rel];
        const int ci = p->types[r.row].c, cj = p->types[r.col].c;
        if (ld < cj) SKF_FAIL(SKF_E_INVALID, "ld too small");
        if (p->f64) {
            copy2d(S, ld, r.S.ptr, cj, ci, cj, 8, as_stream(stream));
        } else {
            hipLaunchKernelGGL((cast_kernel<float, double>), dim3(elem_grid((int64_t)ci * cj)), dim3(256), 0,
                               as_stream(stream), (float*)S, ld, (const double*)r.S.ptr, (int64_t)cj, (int64_t)ci,
                               (int64_t)cj);
            check_launch("cast");
        }
    });
}

int skf_iterate(skf_plan* p, int32_t n_iters, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (p->sliced) SKF_FAIL(SKF_E_STATE, "a plan with row blocks iterates through skf_stage + the exchanges");
        if (n_iters < 0) SKF_FAIL(SKF_E_INVALID, "n_iters < 0");
        for (size_t i = 0; i < p->types.size(); ++i)
            if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "factor of object type %zu not set", i);
        hipStream_t st = as_stream(stream);
        if (p->variant == SKF_TRANSFORM) {
            for (size_t r = 0; r < p->rels.size(); ++r)
                if (!p->rels[r].s_set) SKF_FAIL(SKF_E_STATE, "backbone of relation %zu not set", r);
            if (!p->prepared) prepare_transform(p, st);
            for (int it = 0; it < n_iters; ++it) iterate_transform(p, st);
        } else {
            int it = 0;
            // The iteration is ~50-80 launches; for small graphs (launch-latency regime) the
            // remaining iterations replay ONE captured hipGraph.  Capture needs a real stream
            // (not the legacy default stream) and is skipped while profiling events are recorded.
            if (use_graph(p, st, n_iters)) {
                if (p->first_iter || !p->graph_exec || p->graph_stream != st) {
                    iterate_fit(p, st);               // eager: first-iteration work, attributes
                    ++it;
                    capture_iteration(p, st);
                }
                if (p->graph_exec)
                    for (; it < n_iters; ++it) SKF_HIP(hipGraphLaunch(p->graph_exec, st));
            }
            for (; it < n_iters; ++it) iterate_fit(p, st);
        }
    });
}

int skf_iterate_batch(skf_plan* const* plans, int32_t n_plans, int32_t n_iters, void* stream) {
    return guarded([&] {
        if (!plans || n_plans < 1 || n_plans > SKF_MAX_BATCH) SKF_FAIL(SKF_E_INVALID, "1 .. %d plans", SKF_MAX_BATCH);
        if (n_iters < 0) SKF_FAIL(SKF_E_INVALID, "n_iters < 0");
        skf_plan* p0 = plans[0];
        if (p0 && p0->variant == SKF_TRANSFORM) {
            // fold-ins of one graph into the models of several restarts (reference dfmf.py:191-199): same new relations, the
            // frozen factors / backbones of each restart in its own plan; one launch per iteration serves all of them
            for (int k = 0; k < n_plans; ++k) {
                skf_plan* p = plans[k];
                check_bound(p);
                if (!fold_fused(p) || p->dtype != p0->dtype || p->types.size() != p0->types.size() || p->rels.size() != p0->rels.size() ||
                    p->target != p0->target)
                    SKF_FAIL(SKF_E_STATE, "plan %d does not batch with plan 0 (fold-in without constraints on the target, same graph and engine required)", k);
                for (size_t i = 0; i < p->types.size(); ++i) {
                    if (p->types[i].n != p0->types[i].n || p->types[i].c != p0->types[i].c)
                        SKF_FAIL(SKF_E_STATE, "plan %d: object type %zu differs from plan 0", k, i);
                    if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "plan %d: factor of object type %zu not set", k, i);
                }
                for (size_t r = 0; r < p->rels.size(); ++r) {
                    if (p->rels[r].row != p0->rels[r].row || p->rels[r].col != p0->rels[r].col)
                        SKF_FAIL(SKF_E_STATE, "plan %d: relation %zu differs from plan 0", k, r);
                    if (!p->rels[r].s_set) SKF_FAIL(SKF_E_STATE, "plan %d: backbone of relation %zu not set", k, r);
                }
                for (int q = 0; q < k; ++q)
                    if (plans[q] == p) SKF_FAIL(SKF_E_INVALID, "plan %d listed twice", k);
            }
            hipStream_t st = as_stream(stream);
            for (int k = 0; k < n_plans; ++k)
                if (!plans[k]->prepared) prepare_transform(plans[k], st);
            fold_steps(plans, n_plans, n_iters, st);
            return;
        }
        for (int k = 0; k < n_plans; ++k) {
            skf_plan* p = plans[k];
            check_bound(p);
            for (size_t i = 0; i < p->types.size(); ++i)
                if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "plan %d: factor of object type %zu not set", k, i);
            // one launch serves every plan (plan 0's grid, LDS and job tables): the small-graph schedule, the same engine and
            // the same graph -- object counts, ranks, relation and constraint structure compared field by field
            bool same = p->small_fused && p->f64 == p0->f64 && p->variant == p0->variant && p->engine == p0->engine &&
                        p->types.size() == p0->types.size() && p->rels.size() == p0->rels.size() &&
                        p->thetas.size() == p0->thetas.size() && p->sm_j1.size() == p0->sm_j1.size() &&
                        p->sm_j3.size() == p0->sm_j3.size();
            for (size_t i = 0; same && i < p->types.size(); ++i)
                same = p->types[i].n == p0->types[i].n && p->types[i].c == p0->types[i].c;
            for (size_t r = 0; same && r < p->rels.size(); ++r)
                same = p->rels[r].row == p0->rels[r].row && p->rels[r].col == p0->rels[r].col;
            for (size_t t = 0; same && t < p->thetas.size(); ++t)
                same = p->thetas[t].type == p0->thetas[t].type && p->thetas[t].sparse == p0->thetas[t].sparse;
            auto same_job = [](const SmJob& a, const SmJob& b) {
                return a.kind == b.kind && a.idx == b.idx && a.r0 == b.r0 && a.nr == b.nr && a.part == b.part && a.k0 == b.k0 && a.nk == b.nk;
            };
            for (size_t j = 0; same && j < p->sm_j1.size(); ++j) same = same_job(p->sm_j1[j], p0->sm_j1[j]);
            for (size_t j = 0; same && j < p->sm_j3.size(); ++j) same = same_job(p->sm_j3[j], p0->sm_j3[j]);
            if (!same)
                SKF_FAIL(SKF_E_STATE, "plan %d does not batch with plan 0 (small-graph schedule, same graph and engine required)", k);
            for (int q = 0; q < k; ++q)
                if (plans[q] == p) SKF_FAIL(SKF_E_INVALID, "plan %d listed twice", k);
        }
        hipStream_t st = as_stream(stream);
        std::vector<const void*> tabs((size_t)n_plans);
        for (int k = 0; k < n_plans; ++k) tabs[(size_t)k] = plans[k]->sm_tables.ptr;
        if (tabs != p0->sm_batch_host) {                  // (the table of tables lives in plan 0's workspace; [0] stays plan 0's own)
            p0->sm_batch_host = tabs;
            SKF_HIP(hipMemcpyAsync(p0->sm_batch.ptr, p0->sm_batch_host.data(), tabs.size() * sizeof(void*), hipMemcpyHostToDevice, st));
        }
        for (int it = 0; it < n_iters; ++it) {
            if (p0->f64) iterate_small_fused_t<double>(p0, st, (unsigned)n_plans);
            else iterate_small_fused_t<float>(p0, st, (unsigned)n_plans);
        }
        for (int k = 0; k < n_plans; ++k) plans[k]->first_iter = false;
    });
}

int skf_plan_batchable(const skf_plan* p, int32_t* yes) {
    return guarded([&] {
        check_bound(p);
        if (!yes) SKF_FAIL(SKF_E_INVALID, "null pointer");
        *yes = (p->small_fused || fold_fused(p)) ? 1 : 0;
    });
}

int skf_small_graph_limits(int32_t* max_rank, int64_t* max_objects, int32_t* max_types, int32_t* max_relations,
                           int32_t* max_constraints, int32_t* constraint_nnz_divisor) {
    return guarded([&] {
        if (max_rank) *max_rank = SMALLC;
        if (max_objects) *max_objects = SM_MAX_OBJECTS;
        if (max_types) *max_types = SM_MAXT;
        if (max_relations) *max_relations = SM_MAXR;
        if (max_constraints) *max_constraints = SM_MAXTH;
        if (constraint_nnz_divisor) *constraint_nnz_divisor = SKF_THETA_SPARSE_DIV;
    });
}

int skf_plan_set_graph(skf_plan* p, int32_t enable) {
    return guarded([&] {
        if (!p) SKF_FAIL(SKF_E_INVALID, "null plan");
        p->graph_on = enable != 0;
    });
}

int skf_accumulate(skf_plan* p, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (p->variant == SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "skf_accumulate: SKF_DFMF / SKF_DFMC plans only");
        if (p->owned) SKF_FAIL(SKF_E_STATE, "a plan with owned rows (SKF_OPT_OWNED_ROWS) iterates through skf_iterate_dist");
        for (size_t i = 0; i < p->types.size(); ++i)
            if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "factor of object type %zu not set", i);
        accumulate_fit(p, as_stream(stream));
    });
}

int skf_apply_update(skf_plan* p, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (p->variant == SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "skf_apply_update: SKF_DFMF / SKF_DFMC plans only");
        apply_update(p, as_stream(stream));
    });
}

int skf_stage(skf_plan* p, int32_t stage, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (p->variant == SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "skf_stage: SKF_DFMF / SKF_DFMC plans only");
        if (p->owned) SKF_FAIL(SKF_E_STATE, "a plan with owned rows (SKF_OPT_OWNED_ROWS) iterates through skf_iterate_dist");
        for (size_t i = 0; i < p->types.size(); ++i)
            if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "factor of object type %zu not set", i);
        hipStream_t st = as_stream(stream);
        switch (stage) {
            case SKF_STAGE_CONTRACT: stage_contract(p, st); break;
            case SKF_STAGE_BACKBONE: stage_backbone(p, st); break;
            case SKF_STAGE_ACCUMULATE: stage_accumulate(p, st); break;
            case SKF_STAGE_UPDATE: apply_update(p, st); break;
            default: SKF_FAIL(SKF_E_INVALID, "unknown stage %d", stage);
        }
    });
}

int skf_exchange_range(const skf_plan* p, int32_t which, size_t* offset, size_t* bytes, int32_t* dtype) {
    return guarded([&] {
        if (!p || !offset || !bytes || !dtype) SKF_FAIL(SKF_E_INVALID, "null argument");
        *dtype = p->mt;
        switch (which) {
            case SKF_X_W: *offset = p->xw_off; *bytes = p->xw_bytes; *dtype = SKF_F64; break;
            case SKF_X_Q: *offset = p->xq_off; *bytes = p->xq_bytes; break;
            case SKF_X_QM: *offset = p->xqm_off; *bytes = p->xqm_bytes; break;
            case SKF_X_ED: *offset = p->acc_off; *bytes = p->acc_bytes; break;
            default: SKF_FAIL(SKF_E_INVALID, "unknown exchange range %d", which);
        }
    });
}

int skf_accumulator_range(const skf_plan* p, size_t* offset, size_t* bytes) {
    return guarded([&] {
        if (!p || !offset || !bytes) SKF_FAIL(SKF_E_INVALID, "null argument");
        *offset = p->acc_off;
        *bytes = p->acc_bytes;
    });
}

int skf_comm_unique_id(void* id128) {
    return guarded([&] {
        if (!id128) SKF_FAIL(SKF_E_INVALID, "null argument");
        NcclUniqueId id;
        const int rc = rccl().get_unique_id(&id);
        if (rc != 0) SKF_FAIL(SKF_E_HIP, "ncclGetUniqueId failed (%d)", rc);
        memcpy(id128, &id, sizeof id);
    });
}

int skf_comm_create(const void* id128, int32_t rank, int32_t world, skf_comm** out) {
    return guarded([&] {
        if (!out || world < 1 || rank < 0 || rank >= world) SKF_FAIL(SKF_E_INVALID, "bad rank / world / pointer");
        skf_comm* c = new skf_comm();
        c->rank = rank; c->world = world;
        if (id128) {
            NcclUniqueId id;
            memcpy(&id, id128, sizeof id);
            const int rc = rccl().comm_init_rank(&c->nccl, world, id, rank);
            if (rc != 0) {
                delete c;
                SKF_FAIL(SKF_E_HIP, "ncclCommInitRank failed: %s", g_rccl.error_string ? g_rccl.error_string(rc) : "?");
            }
        } else if (world != 1) {
            delete c;
            SKF_FAIL(SKF_E_INVALID, "a communicator of %d ranks needs the unique id of skf_comm_unique_id", world);
        }
        *out = c;
    });
}

int skf_comm_create_callback(int32_t rank, int32_t world, skf_collective_fn fn, void* user, skf_comm** out) {
    return guarded([&] {
        if (!out || !fn || world < 1 || rank < 0 || rank >= world) SKF_FAIL(SKF_E_INVALID, "bad rank / world / pointer");
        skf_comm* c = new skf_comm();
        c->rank = rank; c->world = world; c->fn = fn; c->user = user;
        *out = c;
    });
}

int skf_comm_create_null(int32_t rank, int32_t world, skf_comm** out) {
    return guarded([&] {
        if (!out || world < 1 || rank < 0 || rank >= world) SKF_FAIL(SKF_E_INVALID, "bad rank / world / pointer");
        skf_comm* c = new skf_comm();
        c->rank = rank; c->world = world; c->null_comm = true;
        *out = c;
    });
}

int skf_owned_rows(int32_t dtype, int64_t n_obj, int32_t part_index, int32_t part_count, int64_t* begin, int64_t* count,
                   int64_t* chunk) {
    return guarded([&] {
        if (n_obj <= 0 || part_count < 1 || part_index < 0 || part_index >= part_count || !begin || !count || !chunk)
            SKF_FAIL(SKF_E_INVALID, "bad argument");
        if (dtype != SKF_F64 && dtype != SKF_F32 && dtype != SKF_BF16) SKF_FAIL(SKF_E_INVALID, "unknown dtype %d", dtype);
        const int64_t ch = owned_chunk(dtype, n_obj, part_count);
        int64_t lo = ch * part_index, hi = lo + ch;
        if (lo > n_obj) lo = n_obj;
        if (hi > n_obj) hi = n_obj;
        *begin = lo;
        *count = hi - lo;
        *chunk = ch;
    });
}

int skf_abi_version(void) { return SKF_ABI_VERSION; }

int skf_comm_info(const skf_comm* c, int32_t* rank, int32_t* world, int32_t* transport, int32_t* transport_ranks) {
    return guarded([&] {
        if (!c) SKF_FAIL(SKF_E_INVALID, "null communicator");
        if (rank) *rank = c->rank;
        if (world) *world = c->world;
        const int kind = c->nccl ? SKF_COMM_RCCL : c->fn ? SKF_COMM_CALLBACK : c->null_comm ? SKF_COMM_NULL : SKF_COMM_SINGLE;
        if (transport) *transport = kind;
        if (transport_ranks) {
            *transport_ranks = kind == SKF_COMM_SINGLE ? 1 : kind == SKF_COMM_NULL ? 0 : c->world;
            if (c->nccl) {                     // what RCCL itself says about the communicator it built
                int n = -1, r = -1;
                if (g_rccl.comm_count && g_rccl.comm_count(c->nccl, &n) == 0) *transport_ranks = n;
                else *transport_ranks = -1;
                if (g_rccl.comm_user_rank && g_rccl.comm_user_rank(c->nccl, &r) == 0 && r != c->rank)
                    SKF_FAIL(SKF_E_STATE, "RCCL numbers this rank %d, the communicator was created as rank %d", r, c->rank);
            }
        }
    });
}

int skf_launch_count(int64_t* launches) {
    return guarded([&] {
        if (!launches) SKF_FAIL(SKF_E_INVALID, "null pointer");
        *launches = g_launches;
    });
}

int skf_split_clamps(int64_t* clamps) {
    return guarded([&] {
        if (!clamps) SKF_FAIL(SKF_E_INVALID, "null pointer");
        *clamps = g_split_clamps;
    });
}

int skf_comm_destroy(skf_comm* c) {
    return guarded([&] {
        if (!c) return;
        if (c->nccl && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->nccl);
        delete c;
    });
}

int skf_plan_set_comm(skf_plan* p, skf_comm* comm) {
    return guarded([&] {
        if (!p) SKF_FAIL(SKF_E_INVALID, "null plan");
        if (p->variant == SKF_TRANSFORM && comm) SKF_FAIL(SKF_E_INVALID, "skf_plan_set_comm: SKF_DFMF / SKF_DFMC plans only");
        p->comm = comm;
    });
}

int skf_iterate_dist(skf_plan* p, int32_t n_iters, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (p->variant == SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "skf_iterate_dist: SKF_DFMF / SKF_DFMC plans only");
        if (!p->comm) SKF_FAIL(SKF_E_STATE, "no communicator attached to the plan (skf_plan_set_comm)");
        if (n_iters < 0) SKF_FAIL(SKF_E_INVALID, "n_iters < 0");
        for (size_t i = 0; i < p->types.size(); ++i)
            if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "factor of object type %zu not set", i);
        if (p->owned) {
            if (p->comm->world != p->part_count || p->comm->rank != p->part_index)
                SKF_FAIL(SKF_E_STATE, "the communicator is rank %d of %d, the plan part %d of %d", p->comm->rank, p->comm->world,
                         p->part_index, p->part_count);
            for (int it = 0; it < n_iters; ++it) iterate_owned(p, as_stream(stream));
            finalize_owned(p, as_stream(stream));
            return;
        }
        for (int it = 0; it < n_iters; ++it) iterate_dist(p, as_stream(stream));
    });
}

int skf_exchange_bytes(const skf_plan* p, int32_t world, size_t* bytes) {
    return guarded([&] {
        if (!p || !bytes || world < 1) SKF_FAIL(SKF_E_INVALID, "bad argument");
        *bytes = exchange_bytes(p, world);
    });
}

int skf_relation_sqerr(skf_plan* p, int32_t rel, double* out, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (rel < 0 || rel >= (int)p->rels.size() || !out) SKF_FAIL(SKF_E_INVALID, "bad relation index / pointer");
        hipStream_t st = as_stream(stream);
        RelState& r = p->rels[rel];
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        const int ni = (int)r.nr, nj = (int)tj.n, ci = ti.c, cj = tj.c;        // the local rows
        if (r.absent) {
            SKF_HIP(hipMemsetAsync(out, 0, sizeof(double), st));
            return;
        }
        if (r.kn) {
            // The completed relation of _dfmc.py:385-386 is X_o + E (X_o = G_i,prev S_prev G_j,prev^T, E the stored residuals);
            // with X_n = G_i S G_j^T of the current factors
            //     |R_c - X_n|^2 = |X_o - X_n|^2 + sum over the known entries of (r - x_n)^2 - (x_o - x_n)^2 ,  x_o = r - e
            // the first term from c x c Gram / cross-Gram products, the second from one pass over the column lists.
            double* acc = (double*)p->sqpart.ptr;
            auto trace_term = [&](const void* Ai, const void* S1, const void* Bj, const void* S2, double scale, bool first) {
                GemmArgs h = gemm_args(Ai, ci, 1, S1, cj, 1, r.U.ptr, cj, ci, cj, ci, EPI_STORE, 0);          // U = A_i S1
                small_gemm(p, h, st);
                h = gemm_args(r.U.ptr, cj, 1, Bj, cj, 1, r.T1.ptr, cj, ci, cj, cj, EPI_STORE, 0);             // T1 = U B_j
                small_gemm(p, h, st);
                hipLaunchKernelGGL(dot_small_kernel, dim3(1), dim3(256), 0, st, (const double*)S2, (const double*)r.T1.ptr,
                                   (int64_t)ci * cj, scale, acc, first ? 0 : 1);
                check_launch("dot_small");
            };
            auto gram_of = [&](const void* A, const void* B, void* C, int c, int64_t n) {                      // C = A^T B (c x c)
                GemmArgs h = gemm_args(A, 1, c, B, c, 1, C, c, c, c, (int)n, EPI_STORE, 0);
                wide_gemm(p, h, st);
            };
            // (the partial slots [1, waves] belong to the pass below; slot 0 collects the trace terms)
            const void* Gi_b = rows_of(p, ti.G, ti, r.r0);                   // the rows of the block (row ownership: the local ones)
            const void* Gp_b = ti.Gp.ptr ? rows_of(p, ti.Gp, ti, r.r0) : nullptr;
            gram_of(Gi_b, Gi_b, r.Xi.ptr, ci, r.nr);
            gram_of(tj.G.ptr, tj.G.ptr, r.Xj.ptr, cj, tj.n);
            trace_term(r.Xi.ptr, r.S.ptr, r.Xj.ptr, r.S.ptr, 1.0, true);
            if (!p->kn_first) {
                gram_of(Gp_b, Gi_b, r.Xi.ptr, ci, r.nr);                     // G_i,prev^T G_i
                gram_of(tj.G.ptr, tj.Gp.ptr, r.Xj.ptr, cj, tj.n);            // G_j^T G_j,prev
                trace_term(r.Xi.ptr, r.S.ptr, r.Xj.ptr, r.Sp.ptr, -2.0, false);
                gram_of(Gp_b, Gp_b, r.Xi.ptr, ci, r.nr);
                gram_of(tj.Gp.ptr, tj.Gp.ptr, r.Xj.ptr, cj, tj.n);
                trace_term(r.Xi.ptr, r.Sp.ptr, r.Xj.ptr, r.Sp.ptr, 1.0, false);
            }
            GemmArgs g2 = gemm_args(tj.G.ptr, cj, 1, r.S.ptr, 1, cj, r.Tm.ptr, ci, nj, ci, cj, EPI_STORE, 0);  // T = G_j S^T
            mixed_gemm(p, g2, st);
            if (p->bf16) launch_to_bf16<float>((uint16_t*)r.FiB.ptr, r.kn_ldf, (const float*)r.Tm.ptr, (int64_t)ci, tj.n, ci, false, st);
            const int waves = known_pass(p, r, true, SRP_ERR, st, 1);       // the pass writes its partials behind slot 0
                                                                            // (capacity checked before it launches)
            hipLaunchKernelGGL((sum_partials_kernel<double>), dim3(1), dim3(256), 0, st, (const double*)p->sqpart.ptr, waves + 1, out);
            check_launch("sum_partials");
            return;
        }
        GemmArgs g = gemm_args(rows_of(p, ti.G, ti, r.r0), ci, 1, r.S.ptr, cj, 1, r.H.ptr, cj, ni, cj, ci, EPI_STORE, 0);
        mixed_gemm(p, g, st);
        if (p->bf16) {
            // one pass over the stored bf16 relation: bf16 H and G_j on the matrix cores, f32 residual
            launch_tile_epilogue(p, r, MODE_SQERR, st);
            hipLaunchKernelGGL((sum_partials_kernel<double>), dim3(1), dim3(256), 0, st, (const double*)p->sqpart.ptr,
                               cdiv(ni, 256) * cdiv(nj, 256), out);
            check_launch("sum_partials");
            return;
        }
        g = gemm_args(r.H.ptr, cj, 1, tj.G.ptr, 1, cj, (void*)r.R, r.ldr, ni, nj, cj, EPI_SQDIFF, 0);
        g.C2 = p->sqpart.ptr;
        // one partial per workgroup of the tile run_gemm picks for THIS product (an all-f64 product of a small
        // relation with 64 <= c_j <= 1024 runs on the deep 32 x 32 tile)
        const TileCfg t = gemm_tile(GemmTypes{p->mt, p->mt, p->mt}, p->engine, g, p->f64, false);
        const int blocks = cdiv(ni, t.bm) * cdiv(nj, t.bn);
        if ((size_t)blocks > p->sq_elems) SKF_FAIL(SKF_E_STATE, "residual partials: %d tiles > %zu slots", blocks, p->sq_elems);
        plan_gemm(p, g, st);
        if (p->f64)
            hipLaunchKernelGGL((sum_partials_kernel<double>), dim3(1), dim3(256), 0, st, (const double*)p->sqpart.ptr,
                               blocks, out);
        else
            hipLaunchKernelGGL((sum_partials_kernel<float>), dim3(1), dim3(256), 0, st, (const float*)p->sqpart.ptr,
                               blocks, out);
        check_launch("sum_partials");
    });
}

int skf_get_contraction(const skf_plan* p, int32_t rel, int32_t which, void* dst, int64_t ld, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (rel < 0 || rel >= (int)p->rels.size() || !dst || which < 0 || which > 2)
            SKF_FAIL(SKF_E_INVALID, "bad relation index / selector / pointer");
        const RelState& r = p->rels[rel];
        const TypeState& ti = p->types[r.row];
        const TypeState& tj = p->types[r.col];
        if (which == 2 && !r.kn) SKF_FAIL(SKF_E_STATE, "relation %d forms P, not P S^T (which = 2 is for known-entries relations)", rel);
        const Slot& src = which == 0 ? r.P : which == 1 ? r.Q : r.A;
        const int64_t rows = which == 1 ? tj.n : r.nr, cols = which == 0 ? tj.c : ti.c;
        if (p->small_fused && which == 1 && r.Q.ptr) {       // the fused small-graph schedule keeps Q as shares: sum them first
            const int64_t total = rows * cols;
            if (p->f64)
                hipLaunchKernelGGL((sum_parts_kernel<double>), dim3(elem_grid(total)), dim3(256), 0, as_stream(stream), (double*)r.Q.ptr,
                                   (const double*)r.SmQ.ptr, total, r.sm_qparts, total);
            else
                hipLaunchKernelGGL((sum_parts_kernel<float>), dim3(elem_grid(total)), dim3(256), 0, as_stream(stream), (float*)r.Q.ptr,
                                   (const float*)r.SmQ.ptr, total, r.sm_qparts, total);
            check_launch("sum_parts");
        }
        if (!src.ptr || rows <= 0) SKF_FAIL(SKF_E_STATE, "relation %d keeps no %s here", rel, which == 0 ? "P" : "Q");
        if (ld < cols) SKF_FAIL(SKF_E_INVALID, "ld too small");
        copy2d(dst, ld, src.ptr, cols, rows, cols, p->esz, as_stream(stream));
    });
}

int skf_plan_set_profiling(skf_plan* p, int32_t enable) {
    return guarded([&] {
        if (!p) SKF_FAIL(SKF_E_INVALID, "null plan");
        p->profiling = enable != 0;
        p->ev_used = 0;
        p->prof_flops = p->prof_bytes = 0.0;
        p->prof_launches = 0;
    });
}

int skf_plan_get_profile(skf_plan* p, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    return guarded([&] {
        if (!p || !total_ms || !launches || !flops || !bytes) SKF_FAIL(SKF_E_INVALID, "null argument");
        double ms = 0.0;
        for (size_t k = 0; k + 1 < p->ev_used; k += 2) {
            SKF_HIP(hipEventSynchronize(p->ev_pool[k + 1]));
            float t = 0.f;
            SKF_HIP(hipEventElapsedTime(&t, p->ev_pool[k], p->ev_pool[k + 1]));
            ms += t;
        }
        *total_ms = ms;
        *launches = p->prof_launches;
        *flops = p->prof_flops;
        *bytes = p->prof_bytes;
        p->ev_used = 0;
        p->prof_flops = p->prof_bytes = 0.0;
        p->prof_launches = 0;
    });
}

// The stand-alone products run on the CALLER's scratch, which may hold fewer slices than the time model asks for: that is
// the caller's choice, not a sizing rule of the library that fell behind -- skf_split_clamps counts the plans' launches only.
struct CallerScratch {
    int64_t keep = g_split_clamps;
    ~CallerScratch() { g_split_clamps = keep; }
};

int skf_gemm(int32_t dtype, int32_t engine, const skf_gemm_desc* d, void* workspace, size_t workspace_bytes,
             void* stream) {
    return guarded([&] {
        CallerScratch scope;
        if (!d || !d->A || !d->B || !d->C) SKF_FAIL(SKF_E_INVALID, "null argument");
        if (dtype != SKF_F64 && dtype != SKF_F32) SKF_FAIL(SKF_E_INVALID, "skf_gemm: dtype must be SKF_F64 / SKF_F32");
        if (d->epi < EPI_STORE || d->epi > EPI_MASKED_STORE) SKF_FAIL(SKF_E_INVALID, "bad epilogue");
        if ((d->epi == EPI_SPLIT_STORE || d->epi == EPI_SPLIT_ACC) && !d->C2) SKF_FAIL(SKF_E_INVALID, "C2 missing");
        if (d->epi == EPI_MASKED_STORE && !d->mask) SKF_FAIL(SKF_E_INVALID, "mask missing");
        if (d->M < 0 || d->N < 0 || d->K < 0) SKF_FAIL(SKF_E_INVALID, "negative dimension");
        GemmArgs g = gemm_args(d->A, d->sa_m, d->sa_k, d->B, d->sb_k, d->sb_n, d->C, d->ldc, d->M, d->N, d->K,
                               d->epi, d->nan_to_num);
        g.C2 = d->C2; g.ldc2 = d->ldc2 ? d->ldc2 : d->ldc;
        g.mask = d->mask; g.ldmask = d->ldmask;
        g.aop = d->aop;
        GemmTypes ty{dtype, d->a_dtype < 0 ? dtype : d->a_dtype, d->b_dtype < 0 ? dtype : d->b_dtype};
        // X^T X (one operand read both ways, plain store): a symmetric product -- the split-K form computes the tiles on / below
        // the diagonal only (GemmArgs::sym; bit for bit the full product)
        static const int sym_on = env_int("SKF_GRAM_SYM", 1) != 0;
        if (sym_on && d->A == d->B && d->M == d->N && d->sa_m == d->sb_n && d->sa_k == d->sb_k && d->epi == EPI_STORE &&
            d->aop == AOP_NONE && ty.a == ty.b)
            g.sym = 1;
        run_gemm(ty, engine, g, d->splits, workspace, workspace_bytes, as_stream(stream));
    });
}

int skf_gemm_bf16(const void* A, int64_t lda, const void* Bt, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N,
                  int32_t Kp, int32_t splits, void* workspace, size_t workspace_bytes, void* stream) {
    return guarded([&] {
        CallerScratch scope;
        if (!A || !Bt || !C || M < 0 || N < 0 || Kp < 0 || ldc < N) SKF_FAIL(SKF_E_INVALID, "bad argument");
        run_gemm_bf16((const uint16_t*)A, lda, (const uint16_t*)Bt, ldb, C, ldc, M, N, Kp, splits, workspace,
                      workspace_bytes, false, as_stream(stream));
    });
}

int skf_gemm_bf16_tn(const void* A, int64_t lda, const void* Bt, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N,
                     int32_t Kp, int32_t splits, void* workspace, size_t workspace_bytes, void* stream) {
    return guarded([&] {
        CallerScratch scope;
        if (!A || !Bt || !C || M < 0 || N < 0 || Kp < 0 || ldc < N) SKF_FAIL(SKF_E_INVALID, "bad argument");
        run_gemm_bf16((const uint16_t*)A, lda, (const uint16_t*)Bt, ldb, C, ldc, M, N, Kp, splits, workspace,
                      workspace_bytes, false, as_stream(stream), true);
    });
}

int skf_gemm_bits(const void* A, int64_t lda_bytes, const void* Bt, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N,
                  int32_t Kp, int32_t transposed, int32_t splits, void* workspace, size_t workspace_bytes, void* stream) {
    return guarded([&] {
        CallerScratch scope;
        if (!A || !Bt || !C || M < 0 || N < 0 || Kp < 0 || ldc < N) SKF_FAIL(SKF_E_INVALID, "bad argument");
        run_gemm_bf16((const uint16_t*)A, lda_bytes, (const uint16_t*)Bt, ldb, C, ldc, M, N, Kp, splits, workspace,
                      workspace_bytes, false, as_stream(stream), transposed != 0, true);
    });
}

int skf_to_bf16(void* dst, int64_t ldd, int32_t src_dtype, const void* src, int64_t lds, int64_t rows, int64_t cols,
                int32_t transpose, void* stream) {
    return guarded([&] {
        if (!dst || !src || rows < 0 || cols < 0) SKF_FAIL(SKF_E_INVALID, "bad argument");
        hipStream_t st = as_stream(stream);
        if (src_dtype == SKF_F64) launch_to_bf16<double>((uint16_t*)dst, ldd, (const double*)src, lds, rows, cols, transpose != 0, st);
        else if (src_dtype == SKF_F32) launch_to_bf16<float>((uint16_t*)dst, ldd, (const float*)src, lds, rows, cols, transpose != 0, st);
        else if (src_dtype == SKF_BF16) launch_to_bf16<uint16_t>((uint16_t*)dst, ldd, (const uint16_t*)src, lds, rows, cols, transpose != 0, st);
        else SKF_FAIL(SKF_E_INVALID, "bad source dtype");
    });
}

int skf_pinv_sym_workspace_bytes(int32_t n, size_t* bytes) {
    return guarded([&] {
        if (n <= 0 || n > EIGH_MAXN - 1 || !bytes) SKF_FAIL(SKF_E_INVALID, "bad order");
        const size_t np = (size_t)(n + 1) / 2 * 2;
        *bytes = align_up(np * np * 8, 256) * 3 + align_up(np * 8, 256) + 512;
        if (n > SWEEP_MAXN) *bytes += defl_scratch_bytes(1, (int64_t)np * np) + 256;
    });
}

int skf_pinv_sym(int32_t dtype, const void* A, int64_t lda, void* K, int64_t ldk, int32_t n, void* ws,
                 size_t ws_bytes, void* stream) {
    return guarded([&] {
        size_t need = 0;
        if (skf_pinv_sym_workspace_bytes(n, &need) != SKF_OK) SKF_FAIL(SKF_E_INVALID, "bad order %d", n);
        if (!A || !K || !ws || ws_bytes < need) SKF_FAIL(SKF_E_WORKSPACE, "pinv workspace too small / null pointer");
        if (dtype != SKF_F64 && dtype != SKF_F32) SKF_FAIL(SKF_E_INVALID, "bad dtype");
        hipStream_t st = as_stream(stream);
        const int np = (n + 1) / 2 * 2;
        const size_t mat = align_up((size_t)np * np * 8, 256);
        char* base = (char*)ws;
        double* eA = (double*)base;
        double* eV = (double*)(base + mat);
        double* eVs = (double*)(base + 2 * mat);
        double* eW = (double*)(base + 3 * mat);
        int* eN = (int*)(base + 3 * mat + align_up((size_t)np * 8, 256));
        int* eNo = eN + 16;
        int* eOk = eN + 32;
        const int total = np * np;
        if (dtype == SKF_F64)
            hipLaunchKernelGGL((eigh_pack_kernel<double>), dim3(elem_grid(total)), dim3(256), 0, st, eA, np,
                               (const double*)A, lda, n, eN, eNo);
        else
            hipLaunchKernelGGL((eigh_pack_kernel<float>), dim3(elem_grid(total)), dim3(256), 0, st, eA, np,
                               (const float*)A, lda, n, eN, eNo);
        check_launch("eigh_pack");
        EighArgs e;
        e.A = eA; e.V = eV; e.Vs = eVs; e.w = eW; e.stride = (int64_t)np * np; e.wstride = np;
        e.n = eN; e.n_orig = eNo; e.chol_ok = eOk; e.max_sweeps = 30;
        const int tot2 = n * n;
        const Switches sw = Switches::read();          // stand-alone operator: no plan to hold them
        // fast path: the blocked sweep for orders 65 .. 256 (writes a contiguous f64 K itself), else Cholesky inverse + unpack
        const bool sweep = dtype == SKF_F64 && ldk == n && sweep_takes(sw, n);
        if (sweep) {
            PinvBatch pb;
            memset(&pb, 0, sizeof pb);
            pb.K[0] = (double*)K; pb.c[0] = n; pb.n_pad[0] = np;
            launch_sweep(sw, e, pb, 1, n, st);
        } else {
            launch_chol(sw, e, 1, np, st);
            if (dtype == SKF_F64)
                hipLaunchKernelGGL((chol_unpack_kernel<double>), dim3(elem_grid(tot2)), dim3(256), 0, st, (double*)K, ldk,
                                   eV, np, n, eOk);
            else
                hipLaunchKernelGGL((chol_unpack_kernel<float>), dim3(elem_grid(tot2)), dim3(256), 0, st, (float*)K, ldk,
                                   eV, np, n, eOk);
            check_launch("chol_unpack");
        }
        if (sweep && defl_multi_takes(sw, n)) {      // orders above 256: the deflation over several workgroups first
            PinvBatch pb;
            memset(&pb, 0, sizeof pb);
            pb.K[0] = (double*)K; pb.c[0] = n; pb.n_pad[0] = np;
            char* scratch = base + align_up(mat * 3 + align_up((size_t)np * 8, 256) + 512, 256);
            launch_deflation_multi(sw, SKF_ENGINE_MFMA, e, pb, 1, n, scratch, st);
        }
        {
            static DeviceOnce once;
            allow_dynamic_lds(once, pchol_pinv_kernel, PCHOL_LDS_BYTES);
        }
        const int lr = np < PCHOL_LDS_R ? np : PCHOL_LDS_R;
        hipLaunchKernelGGL(pchol_pinv_kernel, dim3(1), dim3(EIGH_THREADS), (size_t)lr * (lr + 1) / 2 * 8, st, e, deflation_lo(sw), 1e-7, lr);
        check_launch("pchol_pinv");
        hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(1), dim3(EIGH_THREADS), 0, st, e);
        check_launch("jacobi_eigh");
        if (dtype == SKF_F64)
            hipLaunchKernelGGL((eigh_unpack_pinv_kernel<double>), dim3(elem_grid(tot2)), dim3(256), 0, st, (double*)K,
                               ldk, eVs, eV, np, n, eOk);
        else
            hipLaunchKernelGGL((eigh_unpack_pinv_kernel<float>), dim3(elem_grid(tot2)), dim3(256), 0, st, (float*)K,
                               ldk, eVs, eV, np, n, eOk);
        check_launch("eigh_unpack");
    });
}

int skf_fill_uniform(int32_t dtype, void* dst, int64_t rows, int64_t cols, int64_t ld, uint64_t seed, double scale,
                     double shift, void* stream) {
    return guarded([&] {
        if (!dst || rows < 0 || cols < 0 || ld < cols) SKF_FAIL(SKF_E_INVALID, "bad argument");
        hipStream_t st = as_stream(stream);
        const int grid = elem_grid(rows * cols);
        if (dtype == SKF_F64)
            hipLaunchKernelGGL((fill_uniform_kernel<double>), dim3(grid), dim3(256), 0, st, (double*)dst, rows, cols, ld,
                               seed, scale, shift);
        else if (dtype == SKF_F32)
            hipLaunchKernelGGL((fill_uniform_kernel<float>), dim3(grid), dim3(256), 0, st, (float*)dst, rows, cols, ld,
                               seed, scale, shift);
        else if (dtype == SKF_BF16)
            hipLaunchKernelGGL((fill_uniform_kernel<uint16_t>), dim3(grid), dim3(256), 0, st, (uint16_t*)dst, rows, cols,
                               ld, seed, scale, shift);
        else
            SKF_FAIL(SKF_E_INVALID, "bad dtype");
        check_launch("fill_uniform");
    });
}

int skf_fill_unknown_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes) {
    return guarded([&] {
        if (rows < 0 || cols < 0 || !bytes) SKF_FAIL(SKF_E_INVALID, "bad argument");
        *bytes = (size_t)(2 * rows + 2 * cols + 2) * sizeof(double);
    });
}

int skf_fill_unknown(int32_t dtype, void* data, int64_t ld, int64_t rows, int64_t cols, const uint8_t* mask, int64_t mask_ld,
                     int32_t strategy, double value, void* workspace, size_t workspace_bytes, void* stream) {
    return guarded([&] {
        if (!data || rows < 0 || cols < 0 || ld < cols || (mask && mask_ld < cols)) SKF_FAIL(SKF_E_INVALID, "bad argument");
        if (dtype != SKF_F64 && dtype != SKF_F32) SKF_FAIL(SKF_E_INVALID, "skf_fill_unknown: dtype must be SKF_F64 / SKF_F32");
        if (strategy < FILL_MEAN || strategy > FILL_CONST) SKF_FAIL(SKF_E_INVALID, "unknown fill strategy %d", strategy);
        size_t need = 0;
        skf_fill_unknown_workspace_bytes(rows, cols, &need);
        if (!workspace || workspace_bytes < need) SKF_FAIL(SKF_E_WORKSPACE, "fill workspace too small / null");
        if (rows == 0 || cols == 0) return;
        hipStream_t st = as_stream(stream);
        double* stats = (double*)workspace;
        const int rgrid = (int)((rows + 3) / 4 < 2048 ? ((rows + 3) / 4 > 0 ? (rows + 3) / 4 : 1) : 2048);     // 4 waves per workgroup
        if (dtype == SKF_F64) {
            double* X = (double*)data;
            if (strategy != FILL_CONST) {
                hipLaunchKernelGGL((fill_row_stats_kernel<double>), dim3(rgrid), dim3(256), 0, st, X, ld, rows, cols, mask, mask_ld, stats);
                if (strategy == FILL_COL_MEAN)
                    hipLaunchKernelGGL((fill_col_stats_kernel<double>), dim3(elem_grid(cols)), dim3(256), 0, st, X, ld, rows, cols, mask, mask_ld, stats);
                hipLaunchKernelGGL(fill_total_kernel, dim3(1), dim3(256), 0, st, rows, cols, stats);
            }
            hipLaunchKernelGGL((fill_apply_kernel<double>), dim3(elem_grid(rows * cols)), dim3(256), 0, st, X, ld, rows, cols, mask,
                               mask_ld, stats, strategy, value);
        } else {
            float* X = (float*)data;
            if (strategy != FILL_CONST) {
                hipLaunchKernelGGL((fill_row_stats_kernel<float>), dim3(rgrid), dim3(256), 0, st, X, ld, rows, cols, mask, mask_ld, stats);
                if (strategy == FILL_COL_MEAN)
                    hipLaunchKernelGGL((fill_col_stats_kernel<float>), dim3(elem_grid(cols)), dim3(256), 0, st, X, ld, rows, cols, mask, mask_ld, stats);
                hipLaunchKernelGGL(fill_total_kernel, dim3(1), dim3(256), 0, st, rows, cols, stats);
            }
            hipLaunchKernelGGL((fill_apply_kernel<float>), dim3(elem_grid(rows * cols)), dim3(256), 0, st, X, ld, rows, cols, mask,
                               mask_ld, stats, strategy, value);
        }
        check_launch("fill_unknown");
    });
}

int skf_cast(int32_t dst_dtype, void* dst, int64_t ldd, int32_t src_dtype, const void* src, int64_t lds, int64_t rows,
             int64_t cols, void* stream) {
    return guarded([&] {
        if (!dst || !src || rows < 0 || cols < 0) SKF_FAIL(SKF_E_INVALID, "bad argument");
        hipStream_t st = as_stream(stream);
        const int grid = elem_grid(rows * cols);
        if (dst_dtype == SKF_F32 && src_dtype == SKF_F64)
            hipLaunchKernelGGL((cast_kernel<float, double>), dim3(grid), dim3(256), 0, st, (float*)dst, ldd,
                               (const double*)src, lds, rows, cols);
        else if (dst_dtype == SKF_F64 && src_dtype == SKF_F32)
            hipLaunchKernelGGL((cast_kernel<double, float>), dim3(grid), dim3(256), 0, st, (double*)dst, ldd,
                               (const float*)src, lds, rows, cols);
        else if (dst_dtype == SKF_F32 && src_dtype == SKF_F32)
            hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, st, (float*)dst, ldd,
                               (const float*)src, lds, rows, cols);
        else if (dst_dtype == SKF_F64 && src_dtype == SKF_F64)
            hipLaunchKernelGGL((cast_kernel<double, double>), dim3(grid), dim3(256), 0, st, (double*)dst, ldd,
                               (const double*)src, lds, rows, cols);
        else
            SKF_FAIL(SKF_E_INVALID, "unsupported cast %d -> %d", src_dtype, dst_dtype);
        check_launch("cast");
    });
}

}  // extern "C"
