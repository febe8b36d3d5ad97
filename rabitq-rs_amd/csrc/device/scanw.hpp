// scanw.hpp — k_scanw: the candidate scan with ONE WAVE PER QUERY (gfx950, wave64).
//
// Same semantics, inputs and outputs as k_scan (scan.hpp): search_cluster_v2_batched (src/ivf.rs:1901-2129) over the
// block stream k_select_mfma wrote for one query — accumulate_batch (src/simd.rs:972-1184), compute_batch_distances_u16
// (:2090-2140), lower-bound pruning (src/ivf.rs:2013-2058), ip_packed_ex{2,6}_f32 (src/simd.rs:1835-1915), BinaryHeap
// top-k (src/ivf.rs:904-931, :2116-2126).
//
// Why a second kernel.  k_scan gives a query four waves with fixed roles (three scanners, one replay wave) that hand
// work to each other through LDS queues and workgroup barriers.  In the pruned regime a query is a short chain of
// dependent steps (a handful of tiles, each a memory round trip, a lookup phase and a few refine rounds), so most of the
// time three of the four waves wait for the fourth: 71-74 % of the kernel's resident wave-cycles are parked
// (profiles/r4), and resident wave-cycles are what bounds the pipelined rate (register-file residency).
// Here one wave does everything for its query, in program order:
//   fill   one stream entry per lane: block-level lower bound vs the threshold -> live-block queue (wave-private LDS)
//   tile   two live blocks, lane = vector (half-wave = block): sign codes -> LDS LUT lookups -> fused epilogue;
//          the candidates stay IN REGISTERS (lane j = candidate j, already in stream order), no compaction
//   rounds lazy refine: the next (up to 4; from top_k 64: 8, finished as two halves) candidates whose lower bound is below the
//          current TRUE threshold, 16 lanes per candidate, every code unit of the round requested before the first is decoded;
//          then the reference's exact sequential prune / push / pop loop over them in scalar registers (top_k < 64) or one
//          RankRun merge per half (scan.hpp)
// No barrier, no hand-over, the threshold is never stale inside a tile's replay.  Vector-memory results come back in issue
// order, so the order of the requests is the order of use: a round's code units first, then (once per tile) the next tile's
// block records, which arrive while the round is decoded and replayed.  A query costs one wave for about 1.6 x the time k_scan
// keeps four.  Exactness is k_scan's: the tile prunes with the threshold of its start (T only shrinks: `lb >= T_then` implies
// the reference skipped the candidate too), everything else is replayed in stream order against the running threshold.
// Equal distances: the LAZY-TIE rule (below, at `amb_min`).
#pragma once
#include "scan.hpp"

namespace rbq {

#ifndef RBQ_SCANW_WAVES
#define RBQ_SCANW_WAVES 4     // launch-bounds occupancy target (waves per SIMD) of the RankRun instantiations (top_k >= 64): 128 VGPRs
#endif
#ifndef RBQ_SCANW_WAVES1
#define RBQ_SCANW_WAVES1 5    // ... of the top_k < 64 instantiations: 96 VGPRs (they need 84 at D = 960), and 8 KB of LDS per query at D = 960
#endif
#ifndef RBQ_W_PREF_FIRST
#define RBQ_W_PREF_FIRST 0    // 1: the next tile's records are requested BEFORE the first refine round's codes (else behind them)
#endif
#ifndef RBQ_W_RANKRUN1
#define RBQ_W_RANKRUN1 0        // 1: RankRun (a data-parallel merge of a whole batch) also below top_k 64 (measured: no gain over the sorted run)
#endif
#ifndef RBQ_W_PAIRS_RANK
#define RBQ_W_PAIRS_RANK 1     // ... for the RankRun instantiations (top_k >= 64: five times the refinements of top_k = 10, half the rounds)
#endif
#ifndef RBQ_W_PAIRS
#define RBQ_W_PAIRS 0          // a refine round requests two candidates per 16-lane group (finished as two halves of four)
#endif
#ifndef RBQ_W_FILL_BELOW
#define RBQ_W_FILL_BELOW 4    // fill when fewer live blocks than this are queued (two tiles: one running, one to prefetch)
#endif
#ifndef RBQ_W_WIN0
#define RBQ_W_WIN0 2          // stream entries examined by the first fill step (one tile's worth)
#endif
#ifndef RBQ_W_WIN_GROW
#define RBQ_W_WIN_GROW 4
#endif
// live-block ring and the largest fill window.  60 entries + the 8-word batch list = 512 bytes: at D = 960 / 7 bits a query then needs
// exactly 8 KB of LDS (LUT 3840 + rotated query 3840 + 512), i.e. 20 waves per compute unit = five per SIMD.
constexpr uint32_t kWQueueCap = 60;
constexpr uint32_t kWWinMax = 56;
static_assert(RBQ_W_FILL_BELOW - 1 + kWWinMax <= kWQueueCap, "live queue too small");
// rotated query in LDS: the compile-time-dimension kernels decode exactly D / 16 codes per lane (no zero padding behind D needed)
__host__ __device__ inline uint32_t scanw_qlen(uint32_t D, uint32_t ex_bits, bool compile_time_dim) {
    if (!ex_bits) return 0u;
    return (compile_time_dim && ex_w4(D, ex_bits) <= 4u) ? D : ex_qlen(D, ex_bits);
}
__host__ __device__ inline bool scanw_compile_time_dim(uint32_t D, uint32_t Dc) {
    return D == Dc && (D == 128 || D == 256 || D == 384 || D == 512 || D == 768 || D == 960 || D == 1024 || D == 1536);
}

// LDS carve-up of k_scanw (dynamic only, LUT at byte 0): lut[4Dc] u8 | qrot[scanw_qlen] f32 | queue[kWQueueCap] WorkItem | batch[8] u32
// (lanes of a round's candidates) | TR > 1 only: heap_d[64 TR] f32 | heap_s[64 TR] u32 (RankRun's scatter; final heap-sort of the exact
// heap).  TR = 1: the 512 bytes of queue + batch serve the final heap-sort (both are dead by then).
__host__ __device__ inline size_t scanw_lds_bytes(uint32_t Dc, uint32_t D, uint32_t ex_bits, int tr) {
    return (size_t)Dc * 4 + (size_t)scanw_qlen(D, ex_bits, scanw_compile_time_dim(D, Dc)) * 4 + kWQueueCap * sizeof(WorkItem) + 32 +
           ((tr > 1 || RBQ_W_RANKRUN1) ? (size_t)tr * 64 * 8 : (size_t)0);
}

__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
// number of set bits of a wave-uniform mask below this lane (v_mbcnt: no per-lane 64-bit lane mask to keep in registers)
__device__ __forceinline__ uint32_t mask_rank(unsigned long long m) {
    return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t lane_shfl_u32(uint32_t v, uint32_t src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}
__device__ __forceinline__ float lane_shfl_f32(float v, uint32_t src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __float_as_int(v)));
}

// EX: compile-time ex_bits (0/2/6) when DT != 0; ignored (runtime P.ex_bits) when DT == 0.  TR: registers per lane of the
// top-k (1: top_k <= 63; 2: <= 127; 4: <= 255).  Not served here (k_scan does): MSTG scans, heaps outside the registers, and
// instantiations whose code object reports spilled registers (k_scanw.hip).
template <int DT, int EX, int TR>
__global__ __launch_bounds__(64, (TR == 1 ? RBQ_SCANW_WAVES1 : RBQ_SCANW_WAVES)) void k_scanw(ScanParams P) {
    extern __shared__ __align__(16) unsigned char smraw[];
    const uint32_t Dc = DT ? (uint32_t)DT : P.Dc; // code/LUT dimension (x64)
    const uint32_t D = DT ? (uint32_t)DT : P.D;   // padded_dim (ex codes, rotated query)
    const uint32_t ex_bits = DT ? (uint32_t)EX : P.ex_bits;
    const uint32_t qlen = scanw_qlen(D, ex_bits, DT != 0);
    uint8_t* s_lut = smraw;
    float* s_q = reinterpret_cast<float*>(smraw + (size_t)Dc * 4);
    WorkItem* s_queue = reinterpret_cast<WorkItem*>(s_q + qlen);
    uint32_t* s_b = reinterpret_cast<uint32_t*>(s_queue + kWQueueCap); // [8] lanes of a refine round's candidates, in order
    static_assert(kWQueueCap * sizeof(WorkItem) + 32 == 512, "TR = 1: queue + batch list are the 64-entry heap scratch of the final sort");
    float* heap_d = (TR > 1 || RBQ_W_RANKRUN1) ? reinterpret_cast<float*>(s_b + 8) : reinterpret_cast<float*>(s_queue); // [64 TR]
    uint32_t* heap_s = reinterpret_cast<uint32_t*>(heap_d + 64 * TR);

    const uint32_t q = blockIdx.x, lane = threadIdx.x, half = lane >> 5, l32 = lane & 31u;
    const lds_lut_ptr lut0 = (lds_lut_ptr)(uint32_t)0; // == s_lut
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smraw != 0u) __builtin_trap();
    const uint32_t top_k = P.top_k;
    const size_t stride = (size_t)Dc * 4 + 384;
    const size_t exb = ex_bytes_dev(D, ex_bits);
    const uint32_t nunits = ex_w4(D, ex_bits);

    {
        // LUT and rotated query into LDS: four 16-byte loads per lane in flight, LUT and query together (as two plain loops every
        // iteration waited for its load in front of its ds_write: 8 dependent round trips at D = 960 before the first stream entry)
        const uint4* src = reinterpret_cast<const uint4*>(P.lut + (size_t)q * Dc * 4);
        uint4* dst = reinterpret_cast<uint4*>(s_lut);
        const float4* rs = reinterpret_cast<const float4*>(P.rot + (size_t)q * D);
        float4* rd = reinterpret_cast<float4*>(s_q);
        const uint32_t nl = Dc / 4, nr = ex_bits ? qlen / 4 : 0u;
        constexpr int KU = 4;
        for (uint32_t i0 = lane; i0 < nl || i0 < nr; i0 += 64 * KU) {
            uint4 a[KU];
            float4 b[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const uint32_t i = i0 + 64u * u;
                a[u] = i < nl ? src[i] : make_uint4(0u, 0u, 0u, 0u);
                b[u] = (i < nr && i < D / 4) ? rs[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const uint32_t i = i0 + 64u * u;
                if (i < nl) dst[i] = a[u];
                if (i < nr) rd[i] = b[u];
            }
        }
    }
    const QueryConsts qc = P.consts[q];
    const ProbeInfo* probe = P.probe + (size_t)q * P.nprobe;
    const StreamItem* wl = P.wl + (size_t)q * P.wl_stride;
    const uint32_t ns = P.nstream[q];
    wave_lds_sync();

    uint32_t n_skip_l = 0;                      // per lane (summed at the end)
    uint32_t n_skip_u = 0, n_ext = 0, n_est = 0; // wave-uniform
    uint32_t p_code = 0, p_meta = 0, p_pass = 1, p_ref = 0; // traffic counters of an open profile (P.prof)
    const bool count_skips = P.diag != nullptr;

    struct Meta { float f_add, f_rescale, f_error, g_add, g_err, dotqc; };
    struct TileRegs { WorkItem wi; CodeRegs<DT> cc; Meta m; };
    // lower bound of one candidate as a function of its accumulator value: the exact operation sequence of the epilogue
    // (compute_batch_distances_u16, AVX2 body: only the first op is fused)
    auto lb_of = [&](const Meta& m, float accu_f, float& ip, float& est) -> float {
        ip = fmaf(qc.delta, accu_f, qc.sum_vl);
        const float tt = ip + qc.k1x;
        const float rs = m.f_rescale * tt;
        est = m.f_add + m.g_add;
        est = est + rs;
        const float er = m.f_error * m.g_err;
        return est - er;
    };
    const bool bound_ok = qc.amax <= 65535.0f && !(P.filter && P.diag) && !P.no_block_bound;
    auto lane_prunable = [&](const WorkItem& w, const Meta& m, float T) -> bool {
        float d0, d1;
        const float a = lb_of(m, qc.amin, d0, d1), b = lb_of(m, qc.amax, d0, d1);
        const bool prunable = finite_f(a) && finite_f(b) && fminf(a, b) >= T;
        return (l32 >= (w.rank_nvalid & 63u)) || prunable;
    };
    // requests everything a tile needs of block `half` of the next n queued blocks: first code granules + factor rows + probe row
    auto issue_tile = [&](TileRegs& t, uint32_t qh, uint32_t n) __attribute__((always_inline)) {
        t.wi.gblock = 0; t.wi.rank_nvalid = 0;
        if (half < n) t.wi = s_queue[(qh + half) % kWQueueCap];
        const uint8_t* blk = P.blocks + (size_t)t.wi.gblock * stride;
        if (DT && half < n) load_codes<DT>(t.cc, blk, l32);
        if (half < n) { ++p_meta; if (DT) ++p_code; }
        const float* fac = reinterpret_cast<const float*>(blk + (size_t)Dc * 4);
        const ProbeInfo pi = probe[t.wi.rank_nvalid >> 6];
        t.m.f_add = fac[l32]; t.m.f_rescale = fac[32 + l32]; t.m.f_error = fac[64 + l32];
        t.m.g_add = pi.g_add; t.m.g_err = pi.g_err; t.m.dotqc = pi.dotqc;
    };

    // ---- top-k state (the same registers serve the sorted run / RankRun and, after a distance tie, the exact heap) ----
    constexpr bool kRank = RBQ_W_RANKRUN1 || TR > 1; // RankRun (a data-parallel merge of a whole batch) also for one register per lane
    bool fast = !P.exact_heap; // sorted run (SortedRun / RankRun) until a distance tie shows up
    RegHeap<TR> rh;
    rh.hd = 0; rh.hs = 0u; rh.xd = 0; rh.xs = 0u; rh.len = 0u;
    if (kRank && fast) RankRun<TR>::clear(rh);
    int bag_dk = 0x7f800000; // RankRun: bits of the k-th distance (valid once the run holds top_k entries)
    // LAZY TIES.  On the fast path equal distances do not stop the pass; they are kept in a fixed order and two facts are tracked:
    //   amb_min = the smallest key with which an element ever LEFT the top-k (or was turned away from a full one) while an element of
    //             the same key stayed — the only moment at which the reference's choice depends on the layout of its BinaryHeap;
    //   the final run itself (equal neighbours: into_sorted_vec's order of the two is the heap's).
    // Claim: if the final run has no equal neighbours and its maximum differs from amb_min (or the run is not full), the reference
    // returns exactly this run.  Proof: the MULTISET of keys in the reference's heap never depends on its layout (push adds a key,
    // pop removes a maximum key), so its thresholds, skips and pushes are those of the fast path; its heap and the run can differ
    // only in WHICH of several equal-key elements they hold, i.e. only after an element left while an equal one stayed — with key
    // K >= every key still inside.  Keys inside only shrink afterwards: if the final maximum is below K, every element of key K has
    // left both, and the two hold the same elements again; if it equals K, K = amb_min and the check fires.  With the same elements
    // and no equal neighbours the sorted output is unique.  Otherwise the query is re-run with the BinaryHeap emulation.
    int amb_min = 0x7fffffff;
    auto cur_distk = [&]() -> float {
        if (fast) return rh.len < top_k ? INFINITY : __int_as_float(!kRank ? SortedRun<TR>::kth(rh.hd, rh.xd, rh.len, top_k) : bag_dk);
        return rh.len < top_k ? INFINITY : __int_as_float(rh.d_at(0));
    };

    constexpr uint32_t kNU = ex_w4((uint32_t)(DT ? DT : 16), (uint32_t)(EX ? EX : 2)); // code units per lane and vector
    constexpr bool kDual = (RBQ_W_PAIRS || (RBQ_W_PAIRS_RANK && TR > 1)) && DT != 0 && EX != 0 && kNU <= 3u; // two candidates per 16-lane group (packed FMAs: one query read for both)
    constexpr uint32_t G = kDual ? 8u : 4u; // candidates refined per round
    // the ex-code units of a round are requested into registers and decoded later (compile-time dimensions up to 4 units per
    // lane: D <= 1344 at 6 bits); otherwise they are loaded where they are decoded
    constexpr bool kPre = DT != 0 && EX != 0 && kNU <= 4u;

#ifdef RBQ_WSTAMPS
    unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_fill = 0, st_tile = 0, st_rounds = 0, st_a;
    uint32_t st_ntile = 0, st_nround = 0, st_nlive = 0, st_ncand = 0;
    unsigned long long st_r1 = 0, st_r2 = 0, st_r3 = 0, st_b; // inside a round: collect + permutes | loads + dot + reduce | replay
#define WR0() st_b = __builtin_amdgcn_s_memtime()
#define WRS(acc) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - st_b; st_b = t_; } while (0)
#define WSTAMP_BEGIN() st_a = __builtin_amdgcn_s_memtime()
#define WSTAMP_END(acc) acc += __builtin_amdgcn_s_memtime() - st_a
#else
#define WSTAMP_BEGIN()
#define WSTAMP_END(acc)
#define WR0()
#define WRS(acc)
#endif

  for (;;) { // a second pass only after a distance tie on the sorted fast path
    uint32_t pos = 0;                 // next unexamined stream entry
    uint32_t qhead = 0, qcount = 0;   // live-block ring
    uint32_t win = (uint32_t)RBQ_W_WIN0;
    float T = INFINITY;               // the threshold the tiles prune with (k-th distance when the last tile but one was replayed)
    bool tie = false;

    // fill steps until two tiles' worth of live blocks are queued (or the stream ends)
    auto fill_until = [&]() __attribute__((always_inline)) {
        while (pos < ns && qcount < (uint32_t)RBQ_W_FILL_BELOW) {
            WSTAMP_BEGIN();
            const uint32_t i = pos + lane;
            bool live = lane < win && i < ns;
            StreamItem si;
            si.gblock = 0; si.rank_nvalid = 0; si.lbmin = 0.0f; si.pad = 0;
            if (live) si = wl[i];
            if (live && bound_ok && si.lbmin >= T) { // block_lbmin(): no real vector of the block can pass
                live = false;
                if (count_skips) n_skip_l += si.rank_nvalid & 63u;
            }
            const unsigned long long lm = __ballot(live);
            if (live) {
                WorkItem w;
                w.gblock = si.gblock; w.rank_nvalid = si.rank_nvalid;
                s_queue[(qhead + qcount + mask_rank(lm)) % kWQueueCap] = w;
            }
            qcount += (uint32_t)__popcll(lm);
            pos += win;
            win = win * (uint32_t)RBQ_W_WIN_GROW < kWWinMax ? win * (uint32_t)RBQ_W_WIN_GROW : kWWinMax;
            wave_lds_sync();
            WSTAMP_END(st_fill);
        }
    };

    fill_until();
    TileRegs cur = {};
    uint32_t n = qcount < 2u ? qcount : 2u;
    if (n) issue_tile(cur, qhead, n);
    // candidates of the tile just scanned (lane j = candidate j, stream order = lane order): A_rem = the ones not yet consumed
    unsigned long long A_rem = 0ull;
    float A_lb = 0.0f, A_ip = 0.0f, A_d = 0.0f, A_gadd = 0.0f; // A_d: the 1-bit estimate (the distance itself at ex_bits = 0)
    uint32_t A_slot = 0u;
    const uint32_t g = lane >> 4, gl = lane & 15u;
    while (n) {
        // ------------------------------------------------------------------------------------------------ tile step
        WSTAMP_BEGIN();
        {
                qhead = (qhead + n) % kWQueueCap;
                qcount -= n;
                const uint8_t* blk = P.blocks + (size_t)cur.wi.gblock * stride;
                // per-lane bound with the threshold: the whole wave may be prunable without looking anything up
                const bool live_c = !bound_ok || (__ballot(lane_prunable(cur.wi, cur.m, T)) != ~0ull);
                const uint32_t nvalid = cur.wi.rank_nvalid & 63u;
                A_slot = cur.wi.gblock * 32u + l32;
                bool valid = l32 < nvalid;
                if (valid && P.filter) {
                    const uint32_t id32 = (uint32_t)P.ids[A_slot];
                    valid = ((uint64_t)id32 < P.filter_nbits) && ((P.filter[id32 >> 5] >> (id32 & 31u)) & 1u);
                }
                bool surv = false;
                A_gadd = cur.m.g_add;
                if (live_c) { // wave-uniform
                    if (!DT && half < n) ++p_code;
                    const uint32_t accu = (DT ? lookup_codes<DT>(cur.cc, blk, l32, lut0) : accumulate_block_rt(blk, lut0, l32, Dc)) & 0xffffu;
                    A_lb = lb_of(cur.m, (float)accu, A_ip, A_d);
                    if (!finite_f(A_lb)) {
                        A_lb = P.metric == 0 ? 0.0f : -(cur.m.dotqc + qc.qnorm);
                        // the reference skips iff `lower_bound >= distk` (src/ivf.rs:2054): a NaN bound (NaN query, inner product)
                        // is never skipped.  -inf decides every such test the same way and keeps `lb < T` usable below.
                        if (A_lb != A_lb) A_lb = -INFINITY;
                    }
                    surv = valid && (A_lb < T);
                }
                if (valid && !surv) ++n_skip_l;
                A_rem = __ballot(surv);
#ifdef RBQ_WSTAMPS
                ++st_ntile; if (live_c) ++st_nlive; st_ncand += (uint32_t)__popcll(A_rem);
#endif
        }
        WSTAMP_END(st_tile);
        // ------------------------------------------------------------------------------------------------ refine rounds
        // Exact sequential replay of the reference's prune/push/pop loop in stream order, with LAZY refine: a round takes the next
        // candidates whose lb is below the CURRENT threshold (a superset of the ones the reference evaluates, since the threshold
        // only shrinks), refines them in parallel and replays them against the running threshold.
        unsigned long long mt = 0ull;
        uint32_t ncol = 0, rank = 0;
        auto collect = [&](uint32_t gmax) __attribute__((always_inline)) {
            WR0();
            mt = 0ull; ncol = 0;
            if (!A_rem) return;
            const float distk0 = cur_distk();
            const bool want = ((A_rem >> lane) & 1ull) && A_lb < distk0;
            const unsigned long long m = __ballot(want);
            if (m == 0ull) { n_skip_u += (uint32_t)__popcll(A_rem); A_rem = 0ull; return; } // the rest stays pruned: the threshold never grows
            rank = mask_rank(m);
            const uint32_t Gr = ex_bits ? gmax : 64u;
            const bool take = want && rank < Gr;
            mt = __ballot(take);
            ncol = (uint32_t)__popcll(mt);
            // the stretch this round consumes: everything up to the last taken candidate if more are wanted, else all that is left
            unsigned long long consumed = A_rem;
            if ((uint32_t)__popcll(m) > Gr) consumed = A_rem & ((2ull << (63u - (uint32_t)__builtin_clzll(mt))) - 1ull);
            n_skip_u += (uint32_t)__popcll(consumed) - ncol; // not wanted inside the stretch: lb >= distk0 >= every later threshold
            A_rem &= ~consumed;
#ifdef RBQ_WSTAMPS
            ++st_nround;
#endif
            if (ex_bits) {
                p_ref += ncol;
                if (take) s_b[rank] = lane;
            }
            WRS(st_r1);
        };
        // request: the round's ex codes -> registers (PAIRS: two candidates per 16-lane group, the second one for the round's second half)
        bool has0 = false;
        float ip0 = 0.0f, ga0 = 0.0f, ip1 = 0.0f, ga1 = 0.0f, fa0 = 0.0f, fr0 = 0.0f, fa1 = 0.0f, fr1 = 0.0f;
        uint32_t sl0 = 0, sl1 = 0;
        uint4 ea[kPre ? kNU : 1u], eb[kPre ? kNU : 1u];
        auto request = [&](const bool PAIRS) __attribute__((always_inline)) { // (a constant at each call site)
            wave_lds_sync();
            has0 = g < ncol;
            const uint32_t j0 = has0 ? s_b[g] : 0u;
            const uint32_t j1 = (PAIRS && g + 4u < ncol) ? s_b[g + 4u] : j0;
            // the candidates' slot / ip / g_add move from their lanes to the groups that refine them (every lane takes part in
            // the permutes: they sit outside the group-divergent code below)
            sl0 = lane_shfl_u32(A_slot, j0);
            ip0 = lane_shfl_f32(A_ip, j0); ga0 = lane_shfl_f32(A_gadd, j0);
            if (PAIRS) { sl1 = lane_shfl_u32(A_slot, j1); ip1 = lane_shfl_f32(A_ip, j1); ga1 = lane_shfl_f32(A_gadd, j1); }
            if (has0) { // (group-uniform)
                fa0 = P.f_add_ex[sl0]; fr0 = P.f_rescale_ex[sl0];
                if (PAIRS) { fa1 = P.f_add_ex[sl1]; fr1 = P.f_rescale_ex[sl1]; }
                if (kPre) {
                    const uint4* p0 = reinterpret_cast<const uint4*>(P.ex_codes + (size_t)sl0 * exb) + gl;
#pragma unroll
                    for (int j = 0; j < (int)(kPre ? kNU : 1u); ++j) ea[j] = p0[j * 16];
                    if (PAIRS) {
                        const uint4* p1 = reinterpret_cast<const uint4*>(P.ex_codes + (size_t)sl1 * exb) + gl;
#pragma unroll
                        for (int j = 0; j < (int)(kPre ? kNU : 1u); ++j) eb[j] = p1[j * 16];
                    }
                }
            }
        };
        // finish: decode + distances + replay.  A round of up to 8 candidates is requested in ONE go (one memory round trip) but
        // finished as two halves of four (one candidate per 16-lane group each): the second half is decoded only if one of its
        // candidates is still below the threshold the first half left — and the packed two-candidate decode, which needed fifty
        // more registers than this kernel has, is not needed at all.
        auto finish = [&](const bool PAIRS) __attribute__((always_inline)) { // (a constant at each call site)
#pragma unroll 1
            for (uint32_t ph = 0; ph < (PAIRS ? 2u : 1u); ++ph) {
                WR0();
                // this half's candidates: batch ranks 4 ph .. 4 ph + 3
                const unsigned long long mph = ex_bits ? __ballot(((mt >> lane) & 1ull) && (rank >> 2) == ph) : mt; // (ex_bits = 0: up to 64 per round, no halves)
                if (ph == 1u) {
                    if (mph == 0ull) break;
                    if (ex_bits) {
                        const float dnow = cur_distk();
                        if (__ballot(((mph >> lane) & 1ull) && A_lb < dnow) == 0ull) { // the reference skips every one of them too
                            n_skip_u += (uint32_t)__popcll(mph);
                            break;
                        }
                        if (kPre) {
#pragma unroll
                            for (int j = 0; j < (int)(kPre ? kNU : 1u); ++j) ea[j] = eb[j];
                        }
                        ip0 = ip1; ga0 = ga1; fa0 = fa1; fr0 = fr1; sl0 = sl1;
                        has0 = g + 4u < ncol;
                    }
                }
                float v_d0 = 0.0f;
                if (ex_bits && has0) { // (group-uniform)
                    float sa = 0.0f;
                    if (kPre) sa = ex_dot_one_regs<(EX ? EX : 2), (int)(kPre ? kNU : 1u), (DT ? DT / 16 : 1)>(ea, s_q, gl);
                    else {
                        const uint8_t* ex = P.ex_codes + (size_t)sl0 * exb;
                        if (nunits <= (uint32_t)kExRegUnits) {
                            uint4 u[kExRegUnits];
                            ex_load_all(u, ex, gl, nunits);
                            sa = ex_bits == 6 ? ex_dot_all<6>(u, s_q, gl, nunits) : ex_dot_all<2>(u, s_q, gl, nunits);
                        } else sa = ex_bits == 6 ? ex_dot_units<6>(ex, s_q, gl, nunits) : ex_dot_units<2>(ex, s_q, gl, nunits);
                    }
                    sa = group16_reduce(sa);
                    float tt2 = qc.scale * ip0;
                    tt2 = tt2 + sa;
                    tt2 = tt2 + qc.kbx;
                    const float a = fa0 + ga0;
                    const float mm = fr0 * tt2;
                    v_d0 = a + mm;
                }
                // the refined distances go back to the candidates' lanes: candidate k of the half sits in lane 16 (k & 3) of v_d0;
                // at ex_bits = 0 a candidate's distance is its estimate
                float dj = A_d;
                if (ex_bits) dj = lane_shfl_f32(v_d0, (rank & 3u) << 4);
#if RBQ_WSTAMPS == 2
                asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(dj) : "memory");
#endif
                WRS(st_r2);
                // ---- replay the half's candidates in stream order
                if (fast && kRank) {
                    // the whole batch in one data-parallel step (RankRun, scan.hpp)
                    uint32_t c_skip = 0, c_ext = 0, c_est = 0;
                    int dk = bag_dk;
                    const bool t1 = RankRun<TR>::merge_batch(rh, top_k, mph, __float_as_int(A_lb), __float_as_int(dj), A_slot, lane, count_skips,
                                                             reinterpret_cast<int*>(heap_d), heap_s, c_skip, c_ext, c_est, dk, true, amb_min);
                    bag_dk = dk;
                    n_skip_u += c_skip; n_ext += c_ext; n_est += c_est;
                    tie |= t1;
                } else {
                    // one taken candidate after the other, everything in scalar registers (all of it is wave-uniform)
                    uint32_t len_s = HeapOps::uni(rh.len);
                    int dk = fast ? SortedRun<TR>::kth(rh.hd, rh.xd, len_s, top_k) : 0;
                    unsigned long long todo = mph;
                    if (fast && !count_skips && len_s == top_k) {
                        // One data-parallel test in front of the serial loop: the threshold only falls inside the batch, so a candidate
                        // whose lower bound or whose distance is not below the threshold at the START of the batch is skipped or
                        // rejected by the reference as well (an EQUAL distance stays in: the serial loop must report the tie).
                        const int kk0 = HeapOps::key(dk);
                        const int db = __float_as_int(dj);
                        const bool pass = ((mph >> lane) & 1ull) && A_lb < __int_as_float(dk) && (db & 0x7f800000) != 0x7f800000 && HeapOps::key(db) <= kk0;
                        todo = __ballot(pass);
                    }
                    while (todo) {
                        const uint32_t j = (uint32_t)__builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        const float lb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A_lb), (int)j));
                        const float distk = fast ? __int_as_float(dk) : (rh.len < top_k ? INFINITY : __int_as_float(rh.d_at(0)));
                        if (!(lb < distk)) { ++n_skip_u; continue; }            // `lower_bound >= distk`: skipped
                        ++n_ext;
                        const int dbits = __builtin_amdgcn_readlane(__float_as_int(dj), (int)j);
                        if ((dbits & 0x7f800000) == 0x7f800000) continue;          // non-finite distance: dropped
                        ++n_est;
                        const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)A_slot, (int)j);
                        if (fast) {
                            const int ke = HeapOps::key(dbits);
                            const bool was_full = len_s == top_k;
                            const int kk2 = HeapOps::key(dk);
                            if (was_full) {
                                if (ke > kk2) continue;                             // pushed and popped again: no change
                                if (ke == kk2) { amb_min = amb_min < kk2 ? amb_min : kk2; continue; } // turned away with the maximum's key (lazy tie)
                            }
                            (void)SortedRun<TR>::insert(rh.hd, rh.hs, rh.xd, rh.xs, len_s, dbits, slot, lane); // (equal keys: side by side)
                            len_s = len_s < top_k ? len_s + 1u : len_s; // a full run drops its (new) entry top_k
                            dk = SortedRun<TR>::kth(rh.hd, rh.xd, len_s, top_k);
                            if (was_full && HeapOps::key(dk) == kk2) amb_min = amb_min < kk2 ? amb_min : kk2; // the old maximum left, its twin stays
                        } else {
                            rh.push(dbits, slot);
                            if (rh.len > top_k) rh.pop();
                        }
                    }
                    if (fast) rh.len = len_s;
                }
                WRS(st_r3);
                if (fast && tie) break;
            }
        };
        WSTAMP_BEGIN();
        TileRegs nxt = {};
        uint32_t n_next = 0;
        bool first = true;
#if RBQ_W_PREF_FIRST
        first = false;
        fill_until();
        n_next = qcount < 2u ? qcount : 2u;
        if (n_next) issue_tile(nxt, qhead, n_next);
#endif
        collect(G);
        do {
            if (ncol && ex_bits) request(kDual);
            if (first) {
                // The next tile's records are requested HERE: behind the first round's own loads.  Vector-memory results come back
                // in issue order, so a round whose loads were issued after this prefetch would wait for the (cold) records as well;
                // this way the dot product waits for its codes only, and the records arrive while it and the replay run.
                first = false;
                fill_until();
                n_next = qcount < 2u ? qcount : 2u;
                if (n_next) issue_tile(nxt, qhead, n_next);
            }
            if (ncol) {
                finish(kDual);
                if (fast && tie) break;
                collect(G);
            }
        } while (ncol);
        WSTAMP_END(st_rounds);
        if (fast && tie) break; // leave the pass now: the query is re-run with the exact heap
        T = cur_distk();
        cur = nxt;
        n = n_next;
    }
    if (fast && !tie) { // the final look of the lazy-tie rule
        const uint32_t lenf = HeapOps::uni(rh.len);
        bool eq = false;
        int maxkey = (int)0x80000000;
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            const uint32_t i = (uint32_t)r * 64u + lane;
            const int b = r == 0 ? rh.hd : rh.xd[r];
            const int kv = kRank ? b : HeapOps::key(b);
            int below = __builtin_amdgcn_update_dpp((int)0x80000000, kv, 0x138, 0xf, 0xf, false); // wave_shr:1
            if (r > 0) {
                const int pb = r == 1 ? rh.hd : rh.xd[r - 1];
                const int carry = __builtin_amdgcn_readlane(kRank ? pb : HeapOps::key(pb), 63);
                below = lane == 0 ? carry : below;
            }
            eq |= i >= 1u && i < lenf && kv == below;
            if (lenf && (lenf - 1u) >> 6 == (uint32_t)r) maxkey = __builtin_amdgcn_readlane(kv, (int)((lenf - 1u) & 63u));
        }
        tie = __ballot(eq) != 0ull || (lenf == top_k && maxkey == amb_min);
    }
    if (!(fast && tie)) break;
    if (lane == 0 && P.heap_restarts) atomicAdd(P.heap_restarts, 1u);
    fast = false;
    ++p_pass;
    n_skip_l = 0; n_skip_u = 0; n_ext = 0; n_est = 0;
    rh.hd = 0; rh.hs = 0u; rh.xd = 0; rh.xs = 0u; rh.len = 0u;
    bag_dk = 0x7f800000;
    amb_min = 0x7fffffff;
  }

    // ---- results ----------------------------------------------------------------------------------------------------
    uint32_t len = HeapOps::uni(rh.len);
    if (!fast) { // the exact heap: spill to LDS, into_sorted_vec by one lane
#pragma unroll
        for (int r = 0; r < TR; ++r)
            if ((uint32_t)r * 64u + lane < len) {
                heap_d[r * 64 + lane] = __int_as_float(r == 0 ? rh.hd : rh.xd[r]);
                heap_s[r * 64 + lane] = r == 0 ? rh.hs : rh.xs[r];
            }
        wave_lds_sync();
        if (lane == 0) {
            LdsHeap lh{heap_d, heap_s, len};
            lh.into_sorted();
        }
        wave_lds_sync();
    }
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        const uint32_t i = (uint32_t)r * 64u + lane;
        if (i < top_k) {
            uint64_t id = ~0ull;
            float sc = __int_as_float(0x7fc00000);
            if (i < len) {
                float dv;
                uint32_t sl;
                if (fast) {
                    const int b = r == 0 ? rh.hd : rh.xd[r];
                    dv = __int_as_float(!kRank ? b : HeapOps::key(b));
                    sl = r == 0 ? rh.hs : rh.xs[r];
                } else { dv = heap_d[i]; sl = heap_s[i]; }
                id = P.ids[sl];
                sc = P.metric == 0 ? dv : -dv;
            }
            P.out_ids[(size_t)q * top_k + i] = id;
            P.out_scores[(size_t)q * top_k + i] = sc;
        }
    }
    if (P.prof) {
        const uint32_t pc = wave_sum_u32(l32 == 0 ? p_code : 0u), pm = wave_sum_u32(l32 == 0 ? p_meta : 0u);
        if (lane == 0) {
            unsigned long long* pp = P.prof + prof_stripe(q);
            atomicAdd(pp + kProfCodeBlocks, (unsigned long long)pc);
            atomicAdd(pp + kProfMetaBlocks, (unsigned long long)pm);
            atomicAdd(pp + kProfStreamEntries, (unsigned long long)ns * p_pass);
            atomicAdd(pp + kProfQueries, 1ull);
            if (ex_bits) atomicAdd(pp + kProfExEvals, (unsigned long long)p_ref);
        }
    }
    const uint32_t skip_total = (P.diag ? wave_sum_u32(n_skip_l) : 0u) + n_skip_u;
    if (lane == 0) {
        P.out_counts[q] = len;
        if (P.diag) {
#ifdef RBQ_WSTAMPS
            const unsigned long long st_total = __builtin_amdgcn_s_memtime() - st_t0;
#if RBQ_WSTAMPS == 2
            P.diag[(size_t)q * 3 + 0] = (st_r1 & 0xffffffffull) | (st_r2 << 32);
            P.diag[(size_t)q * 3 + 1] = (st_r3 & 0xffffffffull) | (st_rounds << 32);
            P.diag[(size_t)q * 3 + 2] = (unsigned long long)st_ntile | ((unsigned long long)st_nround << 16) | ((unsigned long long)st_nlive << 32) |
                                        ((unsigned long long)st_ncand << 48);
            (void)skip_total; (void)st_total; (void)st_fill; (void)st_tile;
            return;
#endif
            P.diag[(size_t)q * 3 + 0] = (st_total & 0xffffffffull) | (st_fill << 32);
            P.diag[(size_t)q * 3 + 1] = (st_tile & 0xffffffffull) | (st_rounds << 32);
            P.diag[(size_t)q * 3 + 2] = (unsigned long long)st_ntile | ((unsigned long long)st_nround << 16) | ((unsigned long long)st_nlive << 32) |
                                        ((unsigned long long)st_ncand << 48);
            (void)skip_total;
#else
            P.diag[(size_t)q * 3 + 0] = n_est;
            // + the vectors of probed lists that the probe selection proved skipped as a whole (never streamed)
            P.diag[(size_t)q * 3 + 1] = skip_total + (P.dead_skipped ? P.dead_skipped[q] : 0u);
            P.diag[(size_t)q * 3 + 2] = ex_bits ? n_ext : 0;
#endif
        }
    }
}

} // namespace rbq
