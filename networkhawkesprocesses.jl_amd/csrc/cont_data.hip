// Continuous datasets (events, nodes, duration) and device-resident models.
//
// Dataset creation does, once per dataset, everything that does not depend on the
// parameters: validation with the reference's error conditions, the look-back pre-pass
// (first parent of every event's Δtmax window, using the reference's own fp64 test
// `events[j] > t - Δtmax`, src/continuous.jl:291), the node bucketing of child events and
// the partition of buckets into workgroup-sized items.
#include <chrono>
#include <thread>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>

#include <atomic>
#include "nhp_internal.h"

template <typename T>
static nhp_status upload(nhp_ctx *ctx, T **dst, const T *src, size_t n)
{
    *dst = nullptr;
    if (n == 0) n = 1;
    NHP_HIP(ctx, hipMalloc((void **)dst, sizeof(T) * n));
    if (src) NHP_HIP(ctx, hipMemcpyAsync(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice, ctx->stream));
    return NHP_OK;
}

int nhp_pick_group(double kbar)
{
    const char *env = getenv("NHP_GROUP");
    if (env) {
        int g = atoi(env);
        if (g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32 || g == 64) return g;
    }
    // measured on MI355X (tools/sweep3.sh): best width ~ 3*sqrt(mean window), at most 32:
    // 8 at K=8, 16 at K=64, 16-32 at K=512
    const double target = 3.0 * sqrt(kbar > 0.0 ? kbar : 0.0);
    int g = 1;
    while (g < 32 && (double)g * 2.0 <= target) g *= 2;
    return g;
}

extern "C" nhp_status nhp_cont_dataset_create(nhp_ctx *ctx, const double *events, const int64_t *nodes,
                                              int64_t M, int32_t N, double duration, double dt_max,
                                              nhp_cont_dataset **out)
{
    return nhp_cont_dataset_create_columns(ctx, events, nodes, M, N, duration, dt_max, 0, N, out);
}

bool nhp_is_column_shard(const nhp_cont_dataset *ds) { return ds->col_begin != 0 || ds->col_end != ds->N; }

// The log-likelihood is a sum over child nodes c of  -∫λ0_c - Σ_p cnt[p]·W[p,c] + Σ_{i: c_i = c} log λ_i  and the
// gradient is block-separable in the same columns, so one evaluation shards over GPUs by column range: every shard
// holds all events (they are all parents) but builds work items only for its own columns (second way).
extern "C" nhp_status nhp_cont_dataset_create_columns(nhp_ctx *ctx, const double *events, const int64_t *nodes,
                                                      int64_t M, int32_t N, double duration, double dt_max,
                                                      int32_t col_begin, int32_t col_end, nhp_cont_dataset **out)
{
    if (!ctx || !out || M < 0 || N < 1 || (M > 0 && (!events || !nodes))) return NHP_EINVAL;
    *out = nullptr;
    if (col_begin < 0 || col_end > N || col_begin >= col_end) {
        nhp_set_error(ctx, "column range [%d, %d) must be a non-empty part of [0, %d)", col_begin, col_end, N);
        return NHP_EINVAL;
    }
    if (M >= (int64_t)1 << 31) { nhp_set_error(ctx, "n_events must be < 2^31"); return NHP_EINVAL; }
    if (!(duration >= 0.0)) { nhp_set_error(ctx, "duration must be non-negative"); return NHP_EDOMAIN; }
    if (!(dt_max > 0.0)) { nhp_set_error(ctx, "dt_max must be positive"); return NHP_EDOMAIN; }
    for (int64_t i = 0; i < M; ++i) {
        if (nodes[i] < 1 || nodes[i] > N) {
            nhp_set_error(ctx, "node id %lld at event %lld outside 1..%d", (long long)nodes[i], (long long)(i + 1), N);
            return NHP_EDOMAIN;
        }
        if (!(events[i] >= 0.0)) { nhp_set_error(ctx, "time must be non-negative (event %lld)", (long long)(i + 1)); return NHP_EDOMAIN; }
        if (i > 0 && events[i] < events[i - 1]) { nhp_set_error(ctx, "events must be sorted (event %lld)", (long long)(i + 1)); return NHP_EINVAL; }
    }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    // NHP_TIMING=1: where the host side of a dataset's creation spends its time (stderr)
    static const bool timing = getenv("NHP_TIMING") && atoi(getenv("NHP_TIMING")) != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nhp dataset] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    lap("validation");

    nhp_cont_dataset *ds = new nhp_cont_dataset();
    static std::atomic<uint64_t> next_uid{1};
    ds->uid = next_uid.fetch_add(1);
    ds->ctx = ctx; ds->M = M; ds->N = N; ds->duration = duration; ds->dt_max = dt_max;
    ds->col_begin = col_begin; ds->col_end = col_end;
    ds->t_last = M > 0 ? events[M - 1] : 0.0;

    std::vector<int32_t> node32((size_t)M), first((size_t)M);
    ds->h_cnt.assign((size_t)N, 0.0);
    int64_t pairs = 0, f = 0;
    for (int64_t i = 0; i < M; ++i) {
        node32[i] = (int32_t)(nodes[i] - 1);
        ds->h_cnt[node32[i]] += 1.0;
        double thr = events[i] - dt_max;
        while (f < i && !(events[f] > thr)) ++f;
        first[i] = (int32_t)f;
        pairs += i - f;
    }
    ds->pairs = pairs;
    lap("windows");
    // (the crowding statistics of the recursion's truncated-window bound are made at its first use: nhp_dataset_slab_stats)
    for (int64_t i = 0; i < M && events[i] == 0.0; ++i) ds->n_zero_time = i + 1;
    lap("zero-time events");
    for (int64_t i = 0; i < M; ++i) ds->max_window = std::max<int32_t>(ds->max_window, (int32_t)(i - first[i]));
    ds->h_pair_off.assign((size_t)N + 1, 0);
    for (int64_t i = 0; i < M; ++i) ds->h_pair_off[node32[i] + 1] += i - first[i];
    for (int32_t c = 0; c < N; ++c) ds->h_pair_off[c + 1] += ds->h_pair_off[c];
    ds->group = nhp_pick_group(M > 0 ? (double)pairs / (double)M : 0.0);
    lap("pair offsets per node");

    // stable counting sort of children by node
    ds->h_boff.assign((size_t)N + 1, 0);
    for (int64_t i = 0; i < M; ++i) ds->h_boff[node32[i] + 1]++;
    for (int32_t c = 0; c < N; ++c) ds->h_boff[c + 1] += ds->h_boff[c];
    std::vector<nhp_child> child((size_t)M);
    {
        std::vector<int32_t> cur(ds->h_boff.begin(), ds->h_boff.end() - 1);
        for (int64_t i = 0; i < M; ++i) {
            nhp_child &r = child[cur[node32[i]]++];
            r.t = events[i]; r.first = first[i]; r.idx = (int32_t)i;
        }
    }
    // partition buckets into items of at most `chunk` children; every node gets >= 1 item
    // ~1024 items (4 per CU): one item per node when there are that many nodes (a node's column is
    // then staged exactly once), otherwise nodes are cut into runs of M/1024 children
    int64_t chunk = N >= 1024 ? (int64_t)(1.3 * (double)M / (double)N) + 1 : (M + 1023) / 1024;
    chunk = std::max<int64_t>(32, std::min<int64_t>(4096, chunk));
    const char *env = getenv("NHP_CHUNK");
    if (env && atoi(env) > 0) chunk = std::min(atoi(env), 4096);
    std::vector<nhp_item> items;
    // XCD-aware layout: workgroups are dealt round-robin over the 8 XCDs, so item b runs on XCD b % 8.
    // Giving XCD x = s*(8/TP) + g the children of time part s on the nodes with c % (8/TP) == g makes
    // every XCD touch only 1/TP of the event array (its 4 MiB L2 no longer streams all of it) at the
    // price of staging each column TP times.  That price decides: at mean window 8 the staging eats the
    // gain (tools/xcd.sh), at 64 two time parts win 6 %, at 512 four win 12 % (profiles/README.md).
    // NHP_XCD = 0 | 2 | 4 | 8 overrides.  Speed heuristic only: any placement gives the same result.
    const char *xenv = getenv("NHP_XCD");
    const double kbar = M > 0 ? (double)pairs / (double)M : 0.0;
    // Datasets whose pairs are kept as child slices (short and middle windows, below) are STREAMED by their log-likelihood and
    // gradient kernels: no scattered window for an XCD's L2 to hold, so the second staging of every column only costs there
    // (mean window 64: 103.8 us with two time parts, 78.8 us with none; the simulated set 54.5 -> 49.6).
    const int64_t slices_maxk = getenv("NHP_SLICES_MAXK") ? atoi(getenv("NHP_SLICES_MAXK")) : 160;
    const bool slices_on = !(getenv("NHP_SLICES") && atoi(getenv("NHP_SLICES")) == 0);
    const bool sliced = slices_on && pairs > 0 && pairs <= slices_maxk * M && std::isfinite(dt_max) && dt_max > 0.0 && N <= 65534;
    const int TP = xenv ? atoi(xenv) : (kbar >= 192.0 ? 4 : (kbar >= 24.0 && !sliced ? 2 : 0));
    if ((TP == 2 || TP == 4 || TP == 8) && N >= 8 && M >= 16 * (int64_t)N) {
        const int NG = 8 / TP;
        for (int32_t c0 = 0; c0 < N; c0 += NG)
            for (int s = 0; s < TP; ++s)
                for (int g = 0; g < NG; ++g) {
                    nhp_item it;
                    const int32_t c = c0 + g < N ? c0 + g : N - 1;
                    const bool real = c0 + g < N;
                    const int32_t b = ds->h_boff[c], e = ds->h_boff[c + 1];
                    // children of c whose event index lies in [M*s/TP, M*(s+1)/TP): contiguous (time order)
                    auto lower = [&](int64_t bound) {
                        int32_t lo = b, hi = e;
                        while (lo < hi) { int32_t mid = (lo + hi) >> 1; if (child[mid].idx < bound) lo = mid + 1; else hi = mid; }
                        return lo;
                    };
                    it.node = c;
                    it.kbeg = real ? lower(M * s / TP) : b;
                    it.kend = real ? lower(M * (s + 1) / TP) : b;
                    it.first = real && s == 0;
                    items.push_back(it);
                }
    } else
    for (int32_t c = 0; c < N; ++c) {
        int32_t b = ds->h_boff[c], e = ds->h_boff[c + 1];
        int32_t n = e - b;
        int32_t parts = std::max<int32_t>(1, (int32_t)((n + chunk - 1) / chunk));
        for (int32_t q = 0; q < parts; ++q) {
            nhp_item it;
            it.node = c;
            it.kbeg = b + (int32_t)((int64_t)n * q / parts);
            it.kend = b + (int32_t)((int64_t)n * (q + 1) / parts);
            it.first = q == 0;
            items.push_back(it);
        }
    }
    if (col_begin != 0 || col_end != N) {          // a column shard keeps the items of its own columns
        std::vector<nhp_item> own;
        for (const nhp_item &it : items)
            if (it.node >= col_begin && it.node < col_end) own.push_back(it);
        items.swap(own);
    }
    {   // bit 1 of `first`: the node's only item
        std::vector<int32_t> per_node((size_t)N, 0);
        for (const nhp_item &it : items) per_node[(size_t)it.node]++;
        ds->all_sole = !items.empty();
        for (nhp_item &it : items) {
            if (per_node[(size_t)it.node] == 1) it.first |= 2;
            else ds->all_sole = false;
        }
    }
    ds->n_items = (int32_t)items.size();
    for (const nhp_item &it : items) ds->max_item = std::max(ds->max_item, it.kend - it.kbeg);
    lap("bucketing + items");
    // Windowed kernels: children of an item are visited in rounds of (256/G)*U.  Measured orderings
    // (tools/sortcmp.sh, K=8): 2 = whole item by window length (lanes of a wave run equal pair-loop
    // trips; fastest, 43.4 us, FETCH_SIZE 99.9 MB raw), 1 = rounds in time order, window-sorted inside
    // (all workgroups sweep the time axis together: 14 % less L2-miss traffic, 45.4 us), 0 = time
    // order (48.8 us).  The sum does not depend on the order and the order is fixed: deterministic.
    std::vector<nhp_child> child_w(child);
    std::vector<int32_t> wpos((size_t)M);                 // bucket position of the child at each child_w position
    for (int64_t k = 0; k < M; ++k) wpos[(size_t)k] = (int32_t)k;
    {
        const int G = ds->group, U = G <= 8 ? NHP_U_SMALL : (G <= 32 ? NHP_U_MID : 1);
        const int round = (NHP_WBLOCK / G) * U;
        const char *flat = getenv("NHP_SORT");
        const int mode = flat ? atoi(flat) : 2;
        auto longer = [&](int32_t x, int32_t y) { return (child[x].idx - child[x].first) > (child[y].idx - child[y].first); };
        // items are disjoint ranges of wpos: sorted by a few host threads (37 ms of a 230 ms creation on one)
        auto sort_items = [&](size_t b, size_t e) {
            for (size_t q = b; q < e; ++q) {
                const nhp_item &it = items[q];
                if (mode == 2) std::stable_sort(wpos.begin() + it.kbeg, wpos.begin() + it.kend, longer);
                else if (mode == 1)
                    for (int k = it.kbeg; k < it.kend; k += round)
                        std::stable_sort(wpos.begin() + k, wpos.begin() + std::min(k + round, it.kend), longer);
            }
        };
        const size_t nthreads = std::max<size_t>(1, std::min<size_t>(8, std::min<size_t>(std::thread::hardware_concurrency(), items.size() / 64)));
        if (nthreads <= 1 || M < 100000) sort_items(0, items.size());
        else {
            std::vector<std::thread> pool;
            for (size_t w = 0; w < nthreads; ++w)
                pool.emplace_back(sort_items, items.size() * w / nthreads, items.size() * (w + 1) / nthreads);
            for (std::thread &th : pool) th.join();
        }
        for (int64_t k = 0; k < M; ++k) child_w[(size_t)k] = child[(size_t)wpos[(size_t)k]];
    }
    // pair offsets in child_w order (see nhp_cont_dataset::d_poff)
    std::vector<uint32_t> poff;
    const int64_t plist_maxk = getenv("NHP_PLIST_MAXK") ? atoi(getenv("NHP_PLIST_MAXK")) : 40;      // (mean window 32: 95.5 -> 79.8 us; 64: 124 -> 148, so not there)
    if (pairs > 0 && pairs <= plist_maxk * M && pairs < ((int64_t)1 << 31) && std::isfinite(dt_max) && dt_max > 0.0 && N <= 65535) {
        poff.resize((size_t)M + 1);
        uint32_t run = 0;
        for (int64_t k = 0; k < M; ++k) { poff[(size_t)k] = run; run += (uint32_t)(child_w[(size_t)k].idx - child_w[(size_t)k].first); }
        poff[(size_t)M] = run;
    }
    // child slices (nhp_cont_dataset::d_sl_*): rows per slice = its longest window
    std::vector<uint32_t> sl_row;
    std::vector<int32_t> sl_item0;
    // (short AND middle windows: the slices are streamed, 6 bytes a pair, where k_windowed gathers scattered windows --
    //  mean window 64: 115 -> ~70 us; at 512 the 3 GB list would cost what the exponentials do, so not there)
    if (sliced) {
        sl_item0.resize(items.size() + 1);
        uint64_t rows = 0;
        for (size_t q = 0; q < items.size(); ++q) {
            sl_item0[q] = (int32_t)sl_row.size();
            for (int32_t k0 = items[q].kbeg; k0 < items[q].kend; k0 += 64) {
                int32_t longest = 0;
                for (int32_t k = k0; k < std::min(k0 + 64, items[q].kend); ++k)
                    longest = std::max(longest, child_w[(size_t)k].idx - child_w[(size_t)k].first);
                sl_row.push_back((uint32_t)rows);
                rows += (uint64_t)longest;
                ds->sl_max_rows = std::max(ds->sl_max_rows, longest);
            }
        }
        sl_item0[items.size()] = (int32_t)sl_row.size();
        sl_row.push_back((uint32_t)rows);
        // padding must stay a small share (it is read like pairs) and the row index has to fit 32 bits with 64 records a row
        if (rows >= ((uint64_t)1 << 25) || rows * 64 > (uint64_t)pairs * 2 + 4096) { sl_row.clear(); sl_item0.clear(); ds->sl_max_rows = 0; }
        else {
            ds->sl_rows = (int64_t)rows;
            ds->n_slices = (int32_t)sl_row.size() - 1;
            int nb = 0;
            while (((int64_t)1 << nb) <= N) ++nb;                  // bit length of N: the padding records sit on node N
            ds->sl_nb = nb;
            if (timing) fprintf(stderr, "[nhp dataset] child slices: %d slices, %lld rows = %.3f records per pair\n", ds->n_slices, (long long)rows,
                                (double)rows * 64.0 / (double)pairs);
        }
    }
    std::vector<nhp_event> ev((size_t)M);
    lap("window sort + child_w + poff + slices");
    for (int64_t i = 0; i < M; ++i) { ev[i].t = events[i]; ev[i].node = node32[i]; ev[i].pad = 0; }
    // 8-byte records (nhp_internal.h): only where a node fits 16 bits and the span is a finite positive number
    std::vector<uint64_t> ev8;
    if (N <= 65535 && M > 0 && std::isfinite(events[0]) && std::isfinite(events[M - 1]) && events[M - 1] > events[0]) {
        const double t0 = events[0], range = events[M - 1] - t0;
        const int s2 = 47 - std::ilogb(range);                      // range * 2^s2 < 2^48
        if (s2 > -900 && s2 < 900) {
            const double scale = std::ldexp(1.0, s2);
            ev8.resize((size_t)M);
            for (int64_t i = 0; i < M; ++i) {
                double q = std::nearbyint((events[i] - t0) * scale);
                if (q < 0.0) q = 0.0;
                if (q > 281474976710655.0) q = 281474976710655.0;
                ev8[(size_t)i] = ((uint64_t)(uint32_t)node32[i] << 48) | (uint64_t)q;
            }
            ds->ev8_t0 = t0; ds->ev8_scale = scale;
        }
    }

    lap("event records");
    nhp_status s;
    if ((s = upload(ctx, &ds->d_times, events, (size_t)M)) != NHP_OK ||
        (s = upload(ctx, &ds->d_nodes, node32.data(), (size_t)M)) != NHP_OK ||
        (s = upload(ctx, &ds->d_ev, ev.data(), (size_t)M)) != NHP_OK ||
        (!ev8.empty() && (s = upload(ctx, &ds->d_ev8, ev8.data(), (size_t)M)) != NHP_OK) ||
        (!poff.empty() && (s = upload(ctx, &ds->d_poff, poff.data(), (size_t)M + 1)) != NHP_OK) ||
        (!sl_row.empty() && (s = upload(ctx, &ds->d_sl_row, sl_row.data(), sl_row.size())) != NHP_OK) ||
        (!sl_item0.empty() && (s = upload(ctx, &ds->d_sl_item0, sl_item0.data(), sl_item0.size())) != NHP_OK) ||
        (s = upload(ctx, &ds->d_child, child.data(), (size_t)M)) != NHP_OK ||
        (s = upload(ctx, &ds->d_child_w, child_w.data(), (size_t)M)) != NHP_OK ||
        (s = upload(ctx, &ds->d_wpos, wpos.data(), (size_t)M)) != NHP_OK ||
        (s = upload(ctx, &ds->d_boff, ds->h_boff.data(), (size_t)N + 1)) != NHP_OK ||
        (s = upload(ctx, &ds->d_items, items.data(), items.size())) != NHP_OK ||
        (s = upload(ctx, &ds->d_cnt, ds->h_cnt.data(), (size_t)N)) != NHP_OK) {
        nhp_cont_dataset_destroy(ds);
        return s;
    }
    if (hipMalloc((void **)&ds->d_pn, 4 * (size_t)(M ? M : 1)) != hipSuccess) {
        nhp_set_error(ctx, "out of device memory (parent-node buffer)");
        nhp_cont_dataset_destroy(ds);
        return NHP_ENOMEM;
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);   // host vectors go out of scope below
    if (e != hipSuccess) { nhp_set_error(ctx, "upload failed: %s", hipGetErrorString(e)); nhp_cont_dataset_destroy(ds); return NHP_EHIP; }
    lap("uploads");
    *out = ds;
    return NHP_OK;
}

// Crowding of the data for the recursion's truncated-window bound (cont_recursive.hip): h_slab_max[k] = most events inside any
// closed time window of length h_slab_len[k], lengths doubling from the mean gap up to the span.  Data only; made at the
// first call that needs it (it was 164 ms of a 230 ms dataset creation that most callers never use).
nhp_status nhp_dataset_slab_stats(nhp_ctx *ctx, const nhp_cont_dataset *cds)
{
    nhp_cont_dataset *ds = const_cast<nhp_cont_dataset *>(cds);
    if (ds->slab_done) return NHP_OK;
    ds->slab_done = true;
    const int64_t M = ds->M;
    if (M <= 1) return NHP_OK;
    std::vector<double> events((size_t)M);
    NHP_HIP(ctx, hipMemcpyAsync(events.data(), ds->d_times, sizeof(double) * (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!(events[M - 1] > events[0])) return NHP_OK;
    const double span = events[M - 1] - events[0];
    std::vector<double> lens;
    for (double L = span / (double)M; L < span; L *= 2.0) lens.push_back(L);
    std::vector<int64_t> best(lens.size(), 1);
    auto scan = [&](size_t q) {
        const double L = lens[q];
        int64_t b = 1, lo = 0;
        for (int64_t i = 0; i < M; ++i) {
            while (events[i] - events[lo] > L) ++lo;
            b = std::max<int64_t>(b, i - lo + 1);
        }
        best[q] = b;
    };
    const size_t nthreads = std::max<size_t>(1, std::min<size_t>(8, std::thread::hardware_concurrency()));
    if (nthreads <= 1 || M < 100000) { for (size_t q = 0; q < lens.size(); ++q) scan(q); }
    else {
        std::vector<std::thread> pool;
        for (size_t w = 0; w < nthreads; ++w)
            pool.emplace_back([&, w]() { for (size_t q = w; q < lens.size(); q += nthreads) scan(q); });
        for (std::thread &th : pool) th.join();
    }
    ds->h_slab_len = lens;
    ds->h_slab_max = best;
    return NHP_OK;
}

extern "C" void nhp_cont_dataset_destroy(nhp_cont_dataset *ds)
{
    if (!ds) return;
    (void)hipSetDevice(ds->ctx->device);
    (void)hipStreamSynchronize(ds->ctx->stream);
    (void)hipFree(ds->d_times); (void)hipFree(ds->d_nodes); (void)hipFree(ds->d_child); (void)hipFree(ds->d_child_w); (void)hipFree(ds->d_wpos); (void)hipFree(ds->d_ev); (void)hipFree(ds->d_ev8); (void)hipFree(ds->d_poff); (void)hipFree(ds->d_plist); (void)hipFree(ds->d_plq); (void)hipFree(ds->d_pnode);
    (void)hipFree(ds->d_sl_row); (void)hipFree(ds->d_sl_item0); (void)hipFree(ds->d_sl_lo); (void)hipFree(ds->d_sl_hi);
    (void)hipFree(ds->d_sl_L); (void)hipFree(ds->d_sl_Q); (void)hipFree(ds->d_sl_D);
    (void)hipFree(ds->d_ps_row); (void)hipFree(ds->d_ps_perm); (void)hipFree(ds->d_ps_lo); (void)hipFree(ds->d_ps_hi);
    (void)hipFree(ds->d_boff); (void)hipFree(ds->d_items); (void)hipFree(ds->d_cnt); (void)hipFree(ds->d_pn);
    (void)hipFree(ds->d_adj_k); (void)hipFree(ds->d_adj_p); (void)hipFree(ds->d_adj_dt); (void)hipFree(ds->d_adj_lq); (void)hipFree(ds->d_adj_start); (void)hipFree(ds->d_adj_off); (void)hipFree(ds->d_adj_group); (void)hipFree(ds->d_child_cut);
    (void)hipFree(ds->d_rec_ev); (void)hipFree(ds->d_rec_poff); (void)hipFree(ds->d_rec_rank);
    delete ds;
}

extern "C" int64_t nhp_cont_dataset_pairs(const nhp_cont_dataset *ds) { return ds ? ds->pairs : -1; }

// ---- model ------------------------------------------------------------------------------

static nhp_status check_desc(nhp_ctx *ctx, const nhp_cont_model_desc *d)
{
    if (!d || d->n_nodes < 1 || !d->W || !d->lambda0) { nhp_set_error(ctx, "model: missing W / lambda0"); return NHP_EINVAL; }
    if (d->baseline_kind == NHP_BASELINE_LGCP) {
        if (!d->grid_x || d->grid_n < 2) { nhp_set_error(ctx, "LGCP baseline needs a grid of >= 2 points"); return NHP_ESHAPE; }
        if (d->grid_x[0] != 0.0) { nhp_set_error(ctx, "Grid points x must start at 0."); return NHP_EDOMAIN; }
    } else if (d->baseline_kind != NHP_BASELINE_HOMOGENEOUS) {
        return NHP_EINVAL;
    }
    if (d->impulse_kind == NHP_IMPULSE_EXPONENTIAL) {
        if (!d->theta) { nhp_set_error(ctx, "exponential impulse needs theta"); return NHP_EINVAL; }
    } else if (d->impulse_kind == NHP_IMPULSE_LOGITNORMAL) {
        if (!d->mu || !d->tau) { nhp_set_error(ctx, "logit-normal impulse needs mu and tau"); return NHP_EINVAL; }
        if (!(d->dt_max < INFINITY)) { nhp_set_error(ctx, "logit-normal impulse needs a finite dt_max"); return NHP_EDOMAIN; }
    } else {
        return NHP_EINVAL;
    }
    if (!(d->dt_max > 0.0)) { nhp_set_error(ctx, "dt_max must be positive"); return NHP_EDOMAIN; }
    return NHP_OK;
}

static nhp_status copy_params(nhp_ctx *ctx, nhp_cont_model *m, const nhp_cont_model_desc *d)
{
    ++m->version;
    size_t NN = (size_t)m->N * m->N;
    size_t nl = m->baseline_kind == NHP_BASELINE_LGCP ? (size_t)m->N * m->grid_n : (size_t)m->N;
    hipStream_t st = ctx->stream;
    NHP_HIP(ctx, hipMemcpyAsync(m->d_lambda0, d->lambda0, sizeof(double) * nl, hipMemcpyHostToDevice, st));
    if (m->grid_n) {
        NHP_HIP(ctx, hipMemcpyAsync(m->d_grid, d->grid_x, sizeof(double) * m->grid_n, hipMemcpyHostToDevice, st));
        m->grid_end = d->grid_x[m->grid_n - 1];
    }
    if (m->impulse_kind == NHP_IMPULSE_EXPONENTIAL) {
        NHP_HIP(ctx, hipMemcpyAsync(m->d_p1, d->theta, sizeof(double) * NN, hipMemcpyHostToDevice, st));
    } else {
        NHP_HIP(ctx, hipMemcpyAsync(m->d_p1, d->mu, sizeof(double) * NN, hipMemcpyHostToDevice, st));
        NHP_HIP(ctx, hipMemcpyAsync(m->d_p2, d->tau, sizeof(double) * NN, hipMemcpyHostToDevice, st));
    }
    NHP_HIP(ctx, hipMemcpyAsync(m->d_W, d->W, sizeof(double) * NN, hipMemcpyHostToDevice, st));
    if (m->has_A) NHP_HIP(ctx, hipMemcpyAsync(m->d_A, d->A, sizeof(double) * NN, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));   // host pointers are only borrowed for the call
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_model_create(nhp_ctx *ctx, const nhp_cont_model_desc *d, nhp_cont_model **out)
{
    if (!ctx || !out) return NHP_EINVAL;
    *out = nullptr;
    NHP_TRY(check_desc(ctx, d));
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    nhp_cont_model *m = new nhp_cont_model();
    m->ctx = ctx; m->N = d->n_nodes; m->baseline_kind = d->baseline_kind;
    m->grid_n = d->baseline_kind == NHP_BASELINE_LGCP ? d->grid_n : 0;
    m->impulse_kind = d->impulse_kind; m->has_A = d->A != nullptr; m->dt_max = d->dt_max;
    size_t NN = (size_t)m->N * m->N;
    size_t nl = m->grid_n ? (size_t)m->N * m->grid_n : (size_t)m->N;
    nhp_status s = NHP_OK;
    if ((s = upload<double>(ctx, &m->d_lambda0, nullptr, nl)) != NHP_OK ||
        (m->grid_n && (s = upload<double>(ctx, &m->d_grid, nullptr, (size_t)m->grid_n)) != NHP_OK) ||
        (s = upload<double>(ctx, &m->d_p1, nullptr, NN)) != NHP_OK ||
        (m->impulse_kind == NHP_IMPULSE_LOGITNORMAL && (s = upload<double>(ctx, &m->d_p2, nullptr, NN)) != NHP_OK) ||
        (s = upload<double>(ctx, &m->d_W, nullptr, NN)) != NHP_OK ||
        (m->has_A && (s = upload<double>(ctx, &m->d_A, nullptr, NN)) != NHP_OK) ||
        (s = copy_params(ctx, m, d)) != NHP_OK) {
        nhp_cont_model_destroy(m);
        return s;
    }
    *out = m;
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_model_update(nhp_ctx *ctx, nhp_cont_model *m, const nhp_cont_model_desc *d)
{
    if (!ctx || !m) return NHP_EINVAL;
    NHP_TRY(check_desc(ctx, d));
    int grid_n = d->baseline_kind == NHP_BASELINE_LGCP ? d->grid_n : 0;
    if (d->n_nodes != m->N || d->baseline_kind != m->baseline_kind || grid_n != m->grid_n ||
        d->impulse_kind != m->impulse_kind || (d->A != nullptr) != (m->has_A != 0)) {
        nhp_set_error(ctx, "Parameter vector length does not match model parameter length.");
        return NHP_ESHAPE;
    }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    m->dt_max = d->dt_max;
    return copy_params(ctx, m, d);
}

// params!(process, x): [baseline; impulses; weights]  (src/continuous.jl:121-129)
extern "C" nhp_status nhp_cont_model_set_params(nhp_ctx *ctx, nhp_cont_model *m, const double *x, int64_t len)
{
    if (!ctx || !m || !x) return NHP_EINVAL;
    size_t N = (size_t)m->N, NN = N * N;
    const size_t nb = m->baseline_kind == NHP_BASELINE_HOMOGENEOUS ? N : N * (size_t)m->grid_n;   // λ or vcat(λ...)
    size_t nimp = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL ? NN : 2 * NN;
    if ((size_t)len != nb + nimp + NN) {
        nhp_set_error(ctx, "Parameter vector length does not match model parameter length.");
        return NHP_ESHAPE;
    }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    ++m->version;
    hipStream_t st = ctx->stream;
    NHP_HIP(ctx, hipMemcpyAsync(m->d_lambda0, x, sizeof(double) * nb, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(m->d_p1, x + nb, sizeof(double) * NN, hipMemcpyHostToDevice, st));
    if (m->impulse_kind == NHP_IMPULSE_LOGITNORMAL)
        NHP_HIP(ctx, hipMemcpyAsync(m->d_p2, x + nb + NN, sizeof(double) * NN, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(m->d_W, x + nb + nimp, sizeof(double) * NN, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    return NHP_OK;
}

extern "C" void nhp_cont_model_destroy(nhp_cont_model *m)
{
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    (void)hipFree(m->d_mom); (void)hipFree(m->d_rho);
    (void)hipFree(m->d_lambda0); (void)hipFree(m->d_grid); (void)hipFree(m->d_p1);
    (void)hipFree(m->d_p2); (void)hipFree(m->d_W); (void)hipFree(m->d_A);
    delete m;
}

nhp_cont_args nhp_make_args(const nhp_cont_dataset *ds, const nhp_cont_model *m)
{
    nhp_cont_args a;
    const bool ev8_off = getenv("NHP_EV8") && atoi(getenv("NHP_EV8")) == 0;             // (A/B switch, read per call: exact 16-byte records everywhere)
    a.ev8 = ev8_off ? nullptr : ds->d_ev8; a.ev8_t0 = ds->ev8_t0; a.ev8_scale = ds->ev8_scale; a.ev8_inv = ds->ev8_scale > 0.0 ? 1.0 / ds->ev8_scale : 0.0;
    a.poff = ds->d_poff; a.plist = ds->d_plist; a.plq = ds->d_plq; a.pnode = ds->d_pnode;
    a.times = ds->d_times; a.nodes = ds->d_nodes; a.ev = ds->d_ev; a.child = ds->d_child; a.child_w = ds->d_child_w; a.boff = ds->d_boff;
    a.items = ds->d_items; a.cnt = ds->d_cnt;
    a.lambda0 = m->d_lambda0; a.grid = m->d_grid; a.p1 = m->d_p1; a.p2 = m->d_p2; a.W = m->d_W;
    a.A = m->has_A ? m->d_A : nullptr;
    a.col_begin = ds->col_begin; a.col_end = ds->col_end; a.wpos = ds->d_wpos;
    a.M = ds->M; a.N = ds->N; a.grid_n = m->grid_n; a.baseline_kind = m->baseline_kind;
    a.impulse_kind = m->impulse_kind; a.dt_max = ds->dt_max; a.inv_dtmax = 1.0 / ds->dt_max;
    a.duration = ds->duration;
    const char *dbg = getenv("NHP_DBG");
    a.dbg = dbg ? atoi(dbg) : 0;
    return a;
}

nhp_status nhp_check_pair(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m)
{
    if (!ctx || !ds || !m) return NHP_EINVAL;
    if (ds->ctx != ctx || m->ctx != ctx) { nhp_set_error(ctx, "dataset / model belong to another ctx"); return NHP_EINVAL; }
    if (ds->N != m->N) { nhp_set_error(ctx, "dataset has %d nodes, model %d", ds->N, m->N); return NHP_ESHAPE; }
    if (!(ds->dt_max == m->dt_max)) { nhp_set_error(ctx, "dataset dt_max %g != model dt_max %g", ds->dt_max, m->dt_max); return NHP_EINVAL; }
    // LinearInterpolator throws DomainError outside its support (src/utils/interpolation.jl:29)
    if (m->baseline_kind == NHP_BASELINE_LGCP && ds->M > 0 && ds->t_last > m->grid_end) {
        nhp_set_error(ctx, "Value is outside interpolation support (0, %g)", m->grid_end);
        return NHP_EDOMAIN;
    }
    return NHP_OK;
}
