// launch.h -- argument bundles and launcher prototypes shared by the kernel
// translation units and capi.hip.
#pragma once

#include "../../include/diffqcqp_hip.h"
#include "common.h"
#include "tuning.h"

#include <atomic>

#if defined(__HIPCC__)
#include <tuple>
#include <utility>
#endif

namespace dqq {

// Fallback work-list in the caller's workspace (ints).  [0] number of queued
// problem indices, [1] exit ticket of the dense kernel (its last participant
// re-zeroes the header), [2] next unclaimed entry (dynamic pick-up), then 32 sub-tickets one cache line apart,
// entries from [kWsEntries].
constexpr int kWsCount = 0;
constexpr int kWsTicket = 1;
constexpr int kWsNext = 2; // work-list mode with dynamic pick-up: next unclaimed entry
constexpr int kWsRepTop = 8;    // [8..9] bwd_lane_dense.hip, REPORT mode: 64-bit (groups arrived, problems counted); zero between launches
constexpr int kWsFbShadow = 4;  // [4..7]: what this workspace's drain launches last wrote to the feedback buffer, and where (below)
constexpr int kWsFbSkips = 10;    // unchanged reports not sent since the last one that was (worklist_feedback)
constexpr int kWsPerProblem = 11; // bwd_diag.hip: 1 = this list holds single problems of classified mixed tiles (not whole tiles); cleared by the drain
constexpr int kWsDirty = 12;      // sticky: 1 = a kernel found this header inconsistent (see "work-list hygiene" below); dqq_workspace_status reads it, dqq_workspace_reset clears it
constexpr int kWsSubTickets = 32;   // first of 32 sub-tickets, kWsSubStride ints apart
constexpr int kWsSubStride = 32;    // 128 bytes: one sub-ticket per cache line
// N >= 32 (one to sixteen problems per workgroup of the fast kernel: a dense batch through DQQ_P_AUTO queues from
// thousands of workgroups within microseconds): the list is SEGMENTED -- workgroup i appends to segment i mod 32, each
// with its own counter on its own cache line and kWsSegCap(B) slots in the entry area; ws[kWsCount] is a flag there
// (1 = something is queued: an empty list is recognised with one load).  The
// 8192 same-address atomics of a 65536 x 64 batch took 0.12 ms of its forward and 0.10 ms of its backward (round 3).
constexpr int kWsSegCounts = kWsSubTickets + 32 * kWsSubStride; // entries queued on segment g: [kWsSegCounts + g * kWsSubStride]
constexpr int kWsSegNext = kWsSegCounts + 32 * kWsSubStride;   // next unclaimed entry of segment g (dynamic pick-up)
constexpr int kWsEntries = kWsSegNext + 32 * kWsSubStride;
constexpr bool worklist_segmented(int N) { return N >= 32; }
// slots per segment: the workgroups of one residue class hold at most B/32 + 2 * (problems per workgroup <= 256) problems.
// The invariant behind it -- one tile per wave, a grid of exactly ceil(tiles / waves per workgroup) workgroups, at most
// 256 problems per workgroup -- is static_assert'ed where the fast kernels push (fwd_diag.hip, bwd_diag.hip); a
// persistent or grid-stride fast kernel would need another capacity.
DQQ_HD constexpr long kWsSegCap(long B) { return B / 32 + 512; }
// ints behind the header that hold entries: B for the plain list, 32 segments otherwise
DQQ_HD constexpr long kWsEntryInts(long B) { return 32 * kWsSegCap(B); }


// ---- work-list hygiene (round 5).  The protocol rests on an invariant -- "zero-filled once, every call leaves the header
// zeroed" -- that a caller can break: a workspace that was never zeroed, memory scribbled over, a launch chain cut short by an
// error.  The kernels therefore do not TRUST the header:
//   * the fast kernel that fills the list re-zeroes every word only the drain kernel writes that is not zero (exit tickets,
//     pick-up counters: worklist_prepare_begin / _end -- they are idle while it runs, so this is not a race) -- whatever they
//     held is repaired;
//   * a push whose slot would fall outside the entry area is not performed: the caller poisons that tile's outputs with NaN
//     and the header is marked dirty (worklist_push_entries returns false);
//   * a drain kernel clamps the count it reads to the entry area and replaces an entry that is not a problem of this batch
//     by problem 0 (solved once more, to the same values): nothing is read or written out of bounds, no problem that does
//     not exist is "solved", and the header is marked dirty;
//   * stale entries that ARE problems of this batch (a list left behind by an aborted chain) are solved again by the general
//     kernel behind the fast path: the same problem, the right answer -- and the drain re-zeroes the header as always.
// "Dirty" is sticky and host-visible (dqq_workspace_status); dqq_workspace_reset clears everything.
#if defined(__HIPCC__)
// plain list: slots behind the header; segmented list: slots per segment
DQQ_HD constexpr long worklist_capacity(int N, long B) { return worklist_segmented(N) ? kWsSegCap(B) : kWsEntryInts(B); }
// the count in *word (a word of the header ws), clamped to [0, cap]; a negative one is also repaired on the spot (nobody
// would draw the exit tickets of an "empty" list)
static DQQ_D long worklist_checked_count(const int* ws, const int* word, long cap)
{
    const long c = *word;
    if (c < 0 || c > cap) {
        const_cast<int*>(ws)[kWsDirty] = 1;
        if (c < 0) *const_cast<int*>(word) = 0;
        return c < 0 ? 0 : cap;
    }
    return c;
}
static DQQ_D long worklist_checked_entry(const int* ws, long e, long B)
{
    if ((unsigned long)e >= (unsigned long)B) {
        const_cast<int*>(ws)[kWsDirty] = 1;
        return 0;
    }
    return e;
}
// ONE wave of the fast kernel (the first of workgroup 0) LOADS those words when it starts (worklist_prepare_begin: the loads
// are in flight while the wave does its own tile) and looks at them when it is done (worklist_prepare_end): only a word that
// is not zero is written.  On the path every call takes this costs a handful of instructions and no wait; unconditional
// stores to the ~100 cache lines these words sit on delayed that wave -- and with it the end of an 8 us backward -- by
// 0.2 us (A/B of the builds, tools/ab_libs.py).  The drain is launched behind the fast kernel: the end of it is early enough.
struct WorklistIdle {
    int sub, sub_hi, segnext, head;   // lane < 32: its sub-ticket (two words) and segment pick-up; lane 0: ticket | next | report words
};
static DQQ_D WorklistIdle worklist_prepare_begin(const int* __restrict__ ws, int lane)
{
    WorklistIdle w{0, 0, 0, 0};
    if (lane < 32) {
        w.sub = ws[kWsSubTickets + lane * kWsSubStride];
        w.sub_hi = ws[kWsSubTickets + lane * kWsSubStride + 1];   // (the high word of the 64-bit report counters, bwd_lane_dense.hip)
        w.segnext = ws[kWsSegNext + lane * kWsSubStride];
    }
    if (lane == 0) w.head = ws[kWsTicket] | ws[kWsNext] | ws[kWsRepTop] | ws[kWsRepTop + 1];
    return w;
}
static DQQ_D void worklist_prepare_end(int* __restrict__ ws, int lane, const WorklistIdle& w)
{
    if ((w.sub | w.sub_hi | w.segnext | w.head) == 0) return;
    if (lane < 32) {
        ws[kWsSubTickets + lane * kWsSubStride] = 0;
        ws[kWsSubTickets + lane * kWsSubStride + 1] = 0;
        ws[kWsSegNext + lane * kWsSubStride] = 0;
    }
    if (lane == 0) {
        ws[kWsTicket] = 0;
        ws[kWsNext] = 0;
        ws[kWsRepTop] = 0;
        ws[kWsRepTop + 1] = 0;
    }
}
#endif

#if defined(__HIPCC__)
// Work-list mode of the general kernels: the last participant (wave or workgroup) out re-zeroes the
// work-list header for the next call.  Call from ONE lane per participant; `participants` = how many call
// (gridDim.x, or gridDim.x * waves per workgroup).
// With an empty list nothing is touched: hundreds of same-address atomics would otherwise serialise into
// ~13 us of an otherwise empty launch.  With entries, the tickets are drawn in two levels -- participant i on
// sub-ticket i mod 32, the last of each on the top ticket -- so that no address sees more than
// participants / 32 atomics: 4096 tickets on ONE address took ~30 us of the 67 us backward of a dense 65536 x 8
// batch through DQQ_P_AUTO (round 3).
static DQQ_D void worklist_release(int* ws, long count, int participants)
{
    if (count > 0) {
        const int id = (participants == (int)gridDim.x) ? (int)blockIdx.x
                                                        : (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        const int g = id & 31;
        const int members = (participants - g + 31) >> 5;      // ids congruent to g below `participants`
        int* sub = ws + kWsSubTickets + g * kWsSubStride;
        if (atomicAdd(sub, 1) == members - 1) {
            *sub = 0;
            const int groups = participants < 32 ? participants : 32;
            if (atomicAdd(&ws[kWsTicket], 1) == groups - 1) {
                ws[kWsCount] = 0;
                ws[kWsTicket] = 0;
                ws[kWsNext] = 0;
                ws[kWsPerProblem] = 0;
                for (int h = 0; h < 32; ++h) {  // (segmented list, N >= 32)
                    ws[kWsSegCounts + h * kWsSubStride] = 0;
                    ws[kWsSegNext + h * kWsSubStride] = 0;
                }
            }
        }
    }
}
#endif

// ---- the report word and the route hints (round 4; stateless since round 5).  Which kernel drains a work-list best depends
// on how long the list is: the team kernel (bwd_small.hip) for a few thousand problems, the lane-per-problem kernel
// (bwd_lane_dense.hip) when the list fills the chip -- and the host, which picks the kernel, never sees the length (it sits in
// device memory and nothing on this path may wait for the device).  So the CALLER may hand a backward call one 8-byte word
// of host memory the device can write (`report`): the drain launch stores (B, entries it found) there -- when that differs from
// what the same workspace stored there last --, and the caller turns the word into hint flags for its NEXT calls of that kind,
// N and B with dqq_hint_flags(), a pure function: DQQ_F_EXPECT_LONG_LIST (drain with the lane kernel), DQQ_F_EXPECT_DENSE
// (backward: the lane kernel takes the whole batch, no classifying launch; forward of N = 8: one lane per problem).
// A hint, never a dependency: the routes give the same bits on ANY input (tests/test_gpu_parity.py), so a stale, racy or
// wrong hint costs time only.  The library keeps no state: where the word lives, for how long, per which device or stream,
// and whether to hint at all (not under stream capture: a graph is replayed on batches the word knows nothing about) is the
// caller's business -- diffqcqp_amd/_capi.py keeps one word per (device, kind, N).
// The word: bits 0..30 entries found, bit 31 "the entries are single problems" (with the forward's hand-off the fast path
// queues only the non-diagonal problems of a classified tile; otherwise whole tiles of 128 / N), 32..61 B (mod 2^30),
// 62..63 how many times IN A ROW before this one the same workspace reported "three quarters of the batch or more"
// (saturating at 3).
constexpr unsigned long long kFbBMask = 0x3fffffffULL, kFbCountMask = 0x7fffffffULL, kFbPerProblem = 0x80000000ULL;
inline bool hint_applies(int kind, int N) { return (kind == 0 || kind == 1) && N >= 2 && N <= 8 && N % 2 == 0; }
// entries the drain launch that wrote `w` found, if it ran on a batch of B problems; -1: not known.
// *streak (optional): consecutive earlier reports of count >= 3/4 B.
inline long report_count(unsigned long long w, long B, int* streak = nullptr, bool* per_problem = nullptr)
{
    if (streak != nullptr) *streak = 0;
    if (per_problem != nullptr) *per_problem = false;
    if (w == 0 || ((w >> 32) & kFbBMask) != ((unsigned long long)B & kFbBMask)) return -1;
    if (streak != nullptr) *streak = (int)(w >> 62);
    if (per_problem != nullptr) *per_problem = (w & kFbPerProblem) != 0;
    return (long)(w & kFbCountMask);
}
// problems that sit in a 16-problem block with a non-diagonal one (what the fused forward of N = 8 pays for: one pass of its
// general solve per such block), from the word: the count itself when whole tiles were queued, an estimate for scattered
// problems when single problems were
inline long report_count_in_blocks(unsigned long long w, long B)
{
    bool per_problem = false;
    const long c = report_count(w, B, nullptr, &per_problem);
    if (c <= 0 || !per_problem) return c;
    double stay = 1.0 - (double)c / (double)B, p = stay;
    for (int k = 0; k < 4; ++k) p *= p;   // (1 - c/B)^16
    return (long)((double)B * (1.0 - p));
}
#if defined(__HIPCC__)
// Call from ONE lane of the launch.  The store goes to host memory, and a launch that has one in flight ends later
// (headline step +0.5 us, A/B): the workspace header remembers the last word this workspace sent and where, and an unchanged
// word -- every step of a training loop on one kind of batch -- is not sent again.  The streak (above) is what lets the host
// skip the fast path's launch only for a caller whose batches have been all non-diagonal at least twice running: a caller
// that alternates between kinds of batches under one (kind, N, B) never gets there.
static DQQ_D void worklist_feedback(unsigned long long* fb, int* ws, long B, long count)
{
    const bool per_problem = ws[kWsPerProblem] != 0;   // (set by the fast path that filled this list; this launch drains it)
    if (per_problem) ws[kWsPerProblem] = 0;
    if (fb == nullptr) return;
    unsigned long long* shadow = reinterpret_cast<unsigned long long*>(ws + kWsFbShadow);   // (ws: 16-byte aligned)
    const unsigned long long prev = shadow[0], bb = (unsigned long long)B & kFbBMask;
    const bool same_place = shadow[1] == reinterpret_cast<unsigned long long>(fb);
    unsigned long long streak = 0;
    if (same_place && ((prev >> 32) & kFbBMask) == bb && 4 * (long)(prev & kFbCountMask) >= 3 * B && 4 * count >= 3 * B)
        streak = (prev >> 62) < 3 ? (prev >> 62) + 1 : 3;
    const unsigned long long v = (streak << 62) | (bb << 32) | (per_problem ? kFbPerProblem : 0ULL) | (unsigned long long)count;
    // (the shadow is per WORKSPACE, the word may be shared by several -- diffqcqp_amd/_capi.py keeps one per (device, kind,
    // N) --: another workspace's launch may have overwritten a word this one believes unchanged.  Every 64th unchanged
    // report is therefore sent anyway, so that a word can be stale for a bounded number of calls only: ADVICE r5)
    if (prev == v && same_place && ws[kWsFbSkips] < 63) { ws[kWsFbSkips] += 1; return; }
    ws[kWsFbSkips] = 0;
    shadow[0] = v;
    shadow[1] = reinterpret_cast<unsigned long long>(fb);
    __hip_atomic_store(fb, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

#if defined(__HIPCC__)
// Kernel launch that reports ITS OWN status: hipLaunchKernel's return value, not hipGetLastError() -- the
// thread's sticky error slot belongs to the caller (a stale error of theirs is neither returned as ours nor
// cleared).
template <typename... KArgs, typename... Args, size_t... I>
static inline hipError_t launch_impl(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t s,
                                     std::index_sequence<I...>, Args&&... args)
{
    std::tuple<KArgs...> params{static_cast<KArgs>(args)...};
    void* ptrs[] = {static_cast<void*>(&std::get<I>(params))...};
    return hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, s);
}
template <typename... KArgs, typename... Args>
static inline hipError_t launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t s, Args&&... args)
{
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
    return launch_impl(kernel, grid, block, lds, s, std::index_sequence_for<KArgs...>{}, std::forward<Args>(args)...);
}
#endif

#if defined(__HIPCC__)
// Queue the n problems [first, first + n) of the calling wave (n = 0: none) for the general kernel.
// AGG: ONE global atomic per workgroup instead of one per wave -- with N >= 32 a wave tile is 2 to 16 problems, and
// a dense batch through DQQ_P_AUTO otherwise serialises tens of thousands of same-address atomics (0.38 ms at
// B=65536, N=64).  Every wave of the workgroup that has not returned yet must make the call (three workgroup
// barriers; waves that already ended are not waited for).  s_cnt: two ints of LDS.
// SEG: the segmented list (see kWsSegCounts): the counter and the slots of segment blockIdx.x mod 32.
template <bool AGG, bool SEG = false>
static DQQ_D bool worklist_push_entries(int* __restrict__ ws, long B, int n, bool writes, int rank, int entry, int lane,
                                        int* s_cnt)
{
    // n (wave-uniform) entries from this wave; the lanes with `writes` hold them: `entry` goes to slot base + rank.
    // Returns false (wave-uniform) when the slots would fall outside the entry area -- a header that did not start at
    // zero --: nothing is written then, the header is marked dirty and the caller poisons these problems' outputs.
    int* counter = SEG ? ws + kWsSegCounts + (int)(blockIdx.x & 31u) * kWsSubStride : ws + kWsCount;
    int* slots = SEG ? ws + kWsEntries + (long)(blockIdx.x & 31u) * kWsSegCap(B) : ws + kWsEntries;
    const long cap = SEG ? kWsSegCap(B) : kWsEntryInts(B);
    bool ok = true;
    if constexpr (!AGG) {
        if (n > 0) {
            int base = 0;
            if (lane == 0) {
                base = atomicAdd(counter, n);
                if (SEG) ws[kWsCount] = 1;
            }
            base = __shfl(base, 0, 64);
            ok = base >= 0 && (long)base + n <= cap;
            if (ok) { if (writes) slots[base + rank] = entry; }
            else if (lane == 0) ws[kWsDirty] = 1;
        }
    } else {
        if (threadIdx.x == 0) s_cnt[0] = 0;
        __syncthreads();
        int local = 0;
        if (lane == 0 && n > 0) local = atomicAdd(&s_cnt[0], n);
        __syncthreads();
        if (threadIdx.x == 0 && s_cnt[0] > 0) {
            s_cnt[1] = atomicAdd(counter, s_cnt[0]);
            if (SEG) ws[kWsCount] = 1;
        }
        __syncthreads();
        if (n > 0) {
            const int base = s_cnt[1] + __shfl(local, 0, 64);
            ok = s_cnt[1] >= 0 && (long)base + n <= cap;
            if (ok) { if (writes) slots[base + rank] = entry; }
            else if (lane == 0) ws[kWsDirty] = 1;
        }
    }
    return ok;
}
template <bool AGG, bool SEG = false>
static DQQ_D bool worklist_push(int* __restrict__ ws, long B, long first, int n, int lane, int* s_cnt)
{
    return worklist_push_entries<AGG, SEG>(ws, B, n, lane < n, lane, (int)(first + lane), lane, s_cnt);
}

// Readers of the work-list for the kernels that drain it with a fixed stride (the kernels behind tuning options and
// the global-memory kernels).  Plain list: the count word and entry w.  Segmented list (N >= 32): the 32 segment
// counters are summed / scanned on every call (~6 us: these kernels spend 50 us to milliseconds per problem).  The
// counters do not change while a drain kernel runs (the last participant out re-zeroes them, worklist_release).
static DQQ_D long worklist_count(const int* __restrict__ ws, int N, long B)
{
    if (ws[kWsCount] == 0) return 0;   // (an empty list, either kind: one load)
    if (!worklist_segmented(N)) return worklist_checked_count(ws, ws + kWsCount, kWsEntryInts(B));
    long c = 0;
#pragma unroll
    for (int h = 0; h < 32; ++h) c += worklist_checked_count(ws, ws + kWsSegCounts + h * kWsSubStride, kWsSegCap(B));
    return c;
}
// 0 <= w < worklist_count
static DQQ_D long worklist_entry(const int* __restrict__ ws, int N, long B, long w)
{
    if (!worklist_segmented(N)) return worklist_checked_entry(ws, ws[kWsEntries + w], B);
    long base = 0, at = 0;     // entries before segment g; slot of entry w
    int g = 0;
#pragma unroll
    for (int h = 0; h < 32; ++h) {
        const long c = worklist_checked_count(ws, ws + kWsSegCounts + h * kWsSubStride, kWsSegCap(B));
        if (w >= base && w < base + c) { g = h; at = w - base; }
        base += c;
    }
    return worklist_checked_entry(ws, ws[kWsEntries + g * kWsSegCap(B) + at], B);
}

// Dynamic pick-up for the wave-per-problem kernels (one wave per workgroup; iteration counts differ by 2x between
// problems, a fixed stride would leave the grid waiting for its unluckiest wave).  Every value is wave-uniform.
// Plain list: tickets on ws[kWsNext].  Segmented list: a wave starts on segment blockIdx.x mod 32, draws tickets on
// THAT segment's pick-up word and moves on when it is exhausted; after 32 exhausted segments it is done -- no
// prefix sums, and the tickets are spread over 32 addresses as the pushes were.
// The segmented pick-up can be pipelined (the backward kernels do; without the three ahead_* calls next() claims on
// the spot): while a wave works on a problem its ticket for the next one is in flight
// (ahead_issue right after next(), ahead_entry once the problem's own loads have landed, ahead_done at its end).  A
// ticket and the entry behind it are two dependent round trips to memory, ~3 us that the wave otherwise spends idle
// before every problem -- 7 % of a 42 us backward at N = 64.
struct WorkClaim {
    long count;        // direct mode: B; plain list: entries; segmented list: non-zero iff anything is queued
    int seg, left, c;  // segmented: current segment, segments not yet found exhausted, entries of the current one
    bool listed, segd;
    bool primed;       // segmented: `ahead` holds the wave's next problem (-1: none left)
    bool pipelined;    // ahead_done ran after the last next()
    long ahead;
    int flight;        // per lane: the ticket (stage 1) / the entry (stage 2) in flight
    int stage;         // 0 nothing in flight, 1 ticket, 2 entry
    DQQ_D void open(const int* __restrict__ ws, int use_worklist, int N, long B)
    {
        listed = use_worklist != 0;
        segd = listed && worklist_segmented(N);
        count = listed ? (segd ? (long)(ws[kWsCount] != 0) : worklist_checked_count(ws, ws + kWsCount, kWsEntryInts(B))) : B;
        seg = (int)(blockIdx.x & 31u);
        left = 32;
        c = -1;
        primed = false;
        pipelined = false;
        ahead = -1;
        flight = 0;
        stage = 0;
    }
    DQQ_D long claim_segmented(int* __restrict__ ws, long B)
    {
        while (left > 0) {
            if (c < 0) c = __builtin_amdgcn_readfirstlane((int)worklist_checked_count(ws, ws + kWsSegCounts + seg * kWsSubStride, kWsSegCap(B)));
            if (c > 0) {
                const int t = __builtin_amdgcn_readfirstlane(
                    threadIdx.x == 0 ? atomicAdd(&ws[kWsSegNext + seg * kWsSubStride], 1) : 0);
                if (t >= 0 && t < c)
                    return worklist_checked_entry(ws, (long)__builtin_amdgcn_readfirstlane(ws[kWsEntries + seg * kWsSegCap(B) + t]), B);
            }
            seg = (seg + 1) & 31;
            --left;
            c = -1;
        }
        return -1;
    }
    // the next problem of this wave, -1 = none left.  w: the caller's strided counter (direct mode only).
    // (An empty list is left untouched: nobody would reset its words.)
    DQQ_D long next(int* __restrict__ ws, long B, long w)
    {
        if (!listed) return w < count ? w : -1;
        if (count == 0) return -1;
        if (!segd) {
            const long t = __builtin_amdgcn_readfirstlane(threadIdx.x == 0 ? atomicAdd(&ws[kWsNext], 1) : 0);
            return (t >= 0 && t < count) ? worklist_checked_entry(ws, (long)__builtin_amdgcn_readfirstlane(ws[kWsEntries + t]), B) : -1;
        }
        if (!primed || !pipelined) { // the first problem, or a kernel that does not claim ahead
            ahead = claim_segmented(ws, B);
            primed = true;
        }
        return ahead;
    }
    // the three stages of the claim ahead; no-ops outside the segmented mode.  Call each once per problem, in order.
    DQQ_D void ahead_issue(int* __restrict__ ws)
    {
        stage = 0;
        if (segd && left > 0 && c > 0) {
            flight = threadIdx.x == 0 ? atomicAdd(&ws[kWsSegNext + seg * kWsSubStride], 1) : 0;
            stage = 1;
        }
    }
    DQQ_D void ahead_entry(const int* __restrict__ ws, long B)
    {
        if (stage == 1) {
            const int t = __builtin_amdgcn_readfirstlane(flight);
            if (t >= 0 && t < c) {
                flight = ws[kWsEntries + seg * kWsSegCap(B) + t];
                stage = 2;
            } else { // this segment is exhausted: ahead_done walks on
                seg = (seg + 1) & 31;
                --left;
                c = -1;
                stage = 0;
            }
        }
    }
    DQQ_D void ahead_done(int* __restrict__ ws, long B)
    {
        if (!segd) return;
        ahead = (stage == 2) ? worklist_checked_entry(ws, (long)__builtin_amdgcn_readfirstlane(flight), B) : claim_segmented(ws, B);
        stage = 0;
        pipelined = true;
    }
};
#endif

struct FwdArgs {
    const double* P;
    const double* q;
    const double* l_n; // QCQP: (B,N/2) normal forces; box kinds: (B,N) l_min
    const double* mu;  // QCQP: (B,N/2) friction coefficients; box kinds: (B,N) l_max
    const double* v;   // signed box QP only: (B,N)
    double* x;
    long B;
    int N;
    double eps, mu_prox;
    int max_iter, adaptive, layout;
    int* iters;
    int* ws;
    double* pdiag_out;         // optional (B,N): the diagonal of P, for the backward of the same problems
    unsigned char* flags_out;  // optional (B): 1 = the problem's tile was verified diagonal
    double* scratch = nullptr; // caller's scratch behind the work-list (dqq_scratch_bytes), global-memory kernels only
    bool ref_order = false;    // DQQ_F_REFERENCE_ORDER: 16 < N <= 64 on the reference-order kernels instead of the matrix cores
    int hints = 0;             // DQQ_F_EXPECT_* of the call (routes of identical results)
};

struct BwdArgs {
    const double* P;
    const double* q;
    const double* l_n;
    const double* mu;
    const double* x;
    const double* grad_x;
    double* grad_P;
    double* grad_q;
    double* grad_l_n;
    double* grad_mu;
    const double* pdiag;        // optional: what the forward stored (see FwdArgs)
    const unsigned char* flags; // optional
    double* gamma;   // QCQP only, optional
    double* dgamma;  // QCQP only, optional
    long B;
    int N;
    double epsilon;  // dual-recovery threshold (reference default 1e-10)
    int layout;
    int* ir_steps;
    int* ws;
    double* scratch = nullptr; // see FwdArgs
    bool ref_order = false;    // see FwdArgs
    int hints = 0;             // see FwdArgs
    unsigned long long* report = nullptr;   // optional: where the drain launch stores what it found (device-writable host word)
};

// Which N can solve their non-diagonal tiles inside the fast kernel (no fallback launch: an empty
// work-list launch still costs ~4.6 us behind a 13-35 us kernel).  Chosen at launch from the batch size
// (fwd/bwd_diag_fuses_fallback, with the measurements behind the choice).
constexpr bool fwd_diag_fuses(int N) { return N <= 16; }
constexpr bool bwd_diag_fuses(int N) { return N <= 8; }

// kind: 0 QP, 1 QCQP, 2 box QP, 3 signed box QP (forward only)
constexpr int kKindQP = 0, kKindQCQP = 1, kKindBox = 2, kKindSignedBox = 3;

// diagonal fast paths (fwd_diag.hip, bwd_diag.hip)
bool fwd_diag_supported(int N);
bool fwd_diag_fuses_fallback(int N, long B);
bool bwd_diag_fuses_fallback(int N, long B);
int fwd_diag_default_lpp(int N, long B, int kind);
// fuse_opt: -1 built-in choice, 0 never, 1 whenever instantiated.  *needs_fallback: launch the dense kernel
// in work-list mode behind this one.
hipError_t launch_fwd_diag(int kind, const FwdArgs& a, int lpp, int wpb, int fuse_opt, hipStream_t s,
                           bool* needs_fallback);
// the launchers' own decision, for the dispatcher: true = non-diagonal tiles are solved inside the fast kernel
bool fwd_diag_will_fuse(int N, long B, int layout, int fuse_opt);
bool fwd_diag_takes_dense(int kind, int N, long B); // DQQ_P_DENSE batches solved by the fused kernel's group solve
bool bwd_diag_will_fuse(int kind, int N, long B, int layout, int fuse_opt);
// does the general path have a kernel for this size at all
bool fwd_dense_supported(int kind, int N);
bool bwd_dense_supported(int kind, int N);
bool bwd_diag_supported(int N);
hipError_t launch_bwd_diag(int kind, const BwdArgs& a, int wpb, int fuse_opt, hipStream_t s, bool* needs_fallback);

// general dense kernels (dense.hip).  use_worklist: solve only the problems the
// fast path queued in a.ws, then re-zero the work-list header.
int dense_max_n(int kind); // 0 QP fwd/bwd, 1 QCQP fwd, 2 QCQP bwd
hipError_t launch_fwd_dense(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s);
// lane-per-problem forward for N = 2, 4, 6, 8 (fwd_lane_dense.hip); launch_fwd_dense routes to it
bool fwd_lane_dense_supported(int N);
hipError_t launch_fwd_lane_dense(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s);
// team-per-problem forward for N = 10, 12, 14, 16 (fwd_small.hip); launch_fwd_dense routes to it
bool fwd_small_supported(int N);
hipError_t launch_fwd_small(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s);
// wave-per-problem, register-resident forward for N = 64 (dense_wave64.hip); launch_fwd_dense routes to it
bool fwd_dense_wave64_supported(int N);
hipError_t launch_fwd_dense_wave64(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s);
hipError_t launch_bwd_dense(int kind, const BwdArgs& a, bool use_worklist, hipStream_t s);
// workgroup-per-problem kernels with the matrices in global memory: any N (general_any.hip)
// bytes of scratch those kernels need for (kind, N, B): a slice per workgroup of a grid that depends on (N, B) only
size_t any_scratch_bytes(int kind, bool backward, int N, long B);
bool fwd_needs_any(int kind, int N); // does a call of this size reach them
bool bwd_needs_any(int kind, int N, bool ref_order); // (QCQP 42 < N <= 64 only with DQQ_F_REFERENCE_ORDER)
bool bwd_uses_any(int kind, int N, bool ref_order);
int public_max_n(int kind, bool ref_order);           // dqq_max_n
hipError_t launch_fwd_any(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s);
hipError_t launch_bwd_any(int kind, const BwdArgs& a, bool use_worklist, hipStream_t s);
// lane-per-problem backward for N = 2, 4, 6, 8, QP / QCQP (bwd_lane_dense.hip): whole batches declared dense, or -- when the
// feedback word says the list is long -- the drain launch of a work-list
bool bwd_lane_dense_supported(int kind, int N, long B);
hipError_t launch_bwd_lane_dense(int kind, const BwdArgs& a, int mode, hipStream_t s);
// a DQQ_P_AUTO backward whose every problem was queued last time (feedback word): the lane kernel on the whole batch, reporting
bool bwd_lane_takes_auto_batch(int kind, int N, long B, int hints);
// statically sized team backward for even N <= 16, QP / QCQP (bwd_small.hip); launch_bwd_dense routes to it
bool bwd_small_supported(int kind, int N);
hipError_t launch_bwd_small(int kind, const BwdArgs& a, bool use_worklist, hipStream_t s);
// wave-per-problem, register-resident QP backward for N = 64 (dense_wave64.hip); launch_bwd_dense routes to it
bool bwd_dense_wave64_supported(int kind, int N);
hipError_t launch_bwd_dense_wave64(int kind, const BwdArgs& a, bool use_worklist, hipStream_t s);
// wave-per-problem, register-resident QCQP backward for 16 < N <= 32 (bwd_wave_qcqp.hip); launch_bwd_dense routes to it
bool bwd_wave_qcqp_supported(int kind, int N);
hipError_t launch_bwd_wave_qcqp(const BwdArgs& a, bool use_worklist, hipStream_t s);
// the same for 32 < N <= 64 with the system matrix streamed (bwd_wave_qcqp_big.hip)
bool bwd_wave_qcqp_big_supported(int kind, int N);
hipError_t launch_bwd_wave_qcqp_big(const BwdArgs& a, bool use_worklist, hipStream_t s);

} // namespace dqq
