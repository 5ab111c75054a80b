/* DNNw weight blob -> validated host model + device-oriented packings.
 *
 * Replaces, for the HIP engine, what the reference does in src/parse_lpcnet_weights.c
 * (parse_weights :53-77, find_array_check :84-88, find_idx_check :90-113, the *_init binders
 * :115-221) and in the generated init_lpcnet_model(): same array names, same exact-size checks,
 * same failure condition (-> lpcnet_load_model returns -1).  The register/LDS packings built at
 * the end are new: they exist only because the sample loop keeps GRU-A resident in VGPRs.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lpcnet_engine.h"

typedef struct { const char *name; int size; const void *data; } blob_rec;

static int blob_walk(const unsigned char *p, int len, blob_rec *out, int cap)
{
    int n = 0;
    while (len > 0) {
        int32_t h[5];                               /* magic, version, type, size, block_size */
        if (len < 64) return -1;
        memcpy(h, p, sizeof(h));
        if (h[3] <= 0 || h[4] < h[3] || h[4] > len - 64) return -1;
        if (h[4] & 3) return -1;                    /* records are read as 4-byte words: the padded size keeps them aligned */
        if (p[63] != 0) return -1;                  /* name[43] must terminate the string      */
        if (n == cap) return -1;
        out[n].name = (const char *)(p + 20);
        out[n].size = h[3];
        out[n].data = p + 64;
        n++;
        p += 64 + h[4];
        len -= 64 + h[4];
    }
    return n;
}

static const blob_rec *blob_find(const blob_rec *r, int n, const char *name)
{
    for (int i = 0; i < n; i++) if (!strcmp(r[i].name, name)) return &r[i];
    return NULL;
}

static const void *blob_need(const blob_rec *r, int n, const char *name, size_t bytes)
{
    const blob_rec *e = blob_find(r, n, name);
    return (e && (size_t)e->size == bytes) ? e->data : NULL;
}

/* index stream: per 8-row group {count, pos...}; pos multiple of 4 and pos+3 < nb_in */
static const int *blob_need_idx(const blob_rec *r, int n, const char *name, int nb_in, int nb_out, int *total)
{
    const blob_rec *e = blob_find(r, n, name);
    *total = 0;
    if (!e) return NULL;
    const int *idx = (const int *)e->data;
    int remain = e->size / 4;
    while (remain > 0) {
        int cnt = *idx++;
        if (cnt < 0 || cnt > remain - 1) return NULL;     /* (no cnt + 1: the count comes from an untrusted blob) */
        for (int i = 0; i < cnt; i++) {             /* (order and repeats are the exporter's business: the reference accepts them, and so does the packing) */
            int pos = *idx++;
            if (pos < 0 || pos + 3 >= nb_in || (pos & 3)) return NULL;
        }
        nb_out -= 8;
        remain -= cnt + 1;
        *total += cnt;
    }
    return nb_out == 0 ? (const int *)e->data : NULL;
}

/* ------------------------------------------------------------------------------------------- */
typedef struct { int group, count, first_block; const int *pos; } row_group;

static int cmp_group_desc(const void *a, const void *b)
{
    const row_group *x = (const row_group *)a, *y = (const row_group *)b;
    if (x->count != y->count) return y->count - x->count;
    return x->group - y->group;
}

/* ---- float blobs: dealing with split candidate chains -------------------------------------------------------------------
 * Slots are built separately from the candidate-row groups (6 slots, never mixed with update/reset rows) and from the
 * update/reset groups (12 slots).  A wave takes at most one candidate slot and runs it FIRST: candidate rows start from
 * bias + diag*h, which needs neither the new sample's indices nor the gathered embedding rows, so their items fill the
 * window in which everybody waits for the leader and the gather (WIN items long).  Waves 4..7 do not run GRU-B: they
 * compute the HEAD of their candidate chains (the first `head` blocks of every row) one sample ahead, in GRU-B's shadow,
 * park the partial sums, and only the tail is left for the gather window.  (A row's sum is sequential, but nothing says
 * it has to be formed in one go.)  With GRU-B at ~7.2 k clk and ~370 clk per item beside a GRU-B wave, a head of
 * LPCN_DEAL_EH = 20 items measured best.  Every placement of the candidate slots (at most one per wave: 8!/2! = 20 160) is
 * tried, the update/reset slots follow longest-first to the wave that ends up cheapest, and the placement with the
 * smallest step estimate wins:  T(wave) = max(gather lands, candidate tail done) + update/reset items x clk per item
 * (constants from the in-kernel phase clocks, profiles/r03_phase_clocks.txt). */
typedef struct { int items[LPCN_WAVES], nsl[LPCN_WAVES], nzr[LPCN_WAVES], cand[LPCN_WAVES], zr_items[LPCN_WAVES]; } deal_state;
/* Parameters of one dealing, local to the pack_gru_a() call that fills them (model loads may run concurrently). */
typedef struct {
    int eh;                     /* head length of the early candidate items */
    int hw;                     /* first wave that may carry a candidate head (float: 4 = the waves that never run GRU-B; int8: see pack_gru_a) */
    unsigned hmask;             /* bit w: wave w may carry a candidate head (default: the waves >= hw) */
    int tg, t0a, t0b, ci, cu;   /* deal_wave_cost */
    int tl;                     /* wave 0 leads the streams: its own part of GRU-A starts this much later */
} deal_params;

static int deal_head(const deal_params *dp, int w, int cand)
{
    if (!((dp->hmask >> w) & 1u) || cand <= 0) return 0;
    const int eh = dp->eh < LPCN_EARLY_MAX ? dp->eh : LPCN_EARLY_MAX;
    return cand < eh ? cand : eh;
}

/* Finishing time of a wave's part of GRU-A in the sample step, clk after the tree barrier (fitted to profiles/r03_phase_clocks.txt):
 *  - its update/reset items cannot start before the leader has published the indices and the gathered rows have landed and
 *    been folded into the start values: TG;
 *  - a candidate tail of at least 10 items (the kernel's run-ahead mode) starts at T0 and keeps the wave busy at CI clk per
 *    item (both waves of a SIMD running items) -- a shorter tail waits for the gather like everything else. */
static long deal_wave_cost(const deal_params *dp, const deal_state *d, int w)
{
    const int tail = d->cand[w] - deal_head(dp, w, d->cand[w]);       /* candidate items left for the sample step itself */
    const long after_gather = dp->tg + (w == 0 ? dp->tl : 0) + (long)dp->cu * d->zr_items[w];
    long t;
    if (tail >= 10) {
        t = (!((dp->hmask >> w) & 1u) ? dp->t0a : dp->t0b) + (w == 0 ? dp->tl : 0) + (long)dp->ci * (tail + d->zr_items[w]);
        if (t < after_gather) t = after_gather;
    } else {
        t = after_gather + (long)dp->cu * tail;
    }
    return t + 100L * d->nsl[w];
}

/* slot_max[0..nc) candidate slots, [nc..ns) update/reset slots (both descending); returns 0 and wave_of[] or -1 */
static int deal_v2(const deal_params *dp, const int *slot_max, int nc, int ns, int cap, int *wave_of)
{
    int perm[8], used[LPCN_WAVES] = {0}, best_wave[32];
    long best_max = -1, best_sum = 0;
    if (nc > 6 || ns > 32) return -1;
    /* iterative enumeration of injective maps candidate slot i -> wave perm[i] */
    int depth = 0;
    for (int i = 0; i < 8; i++) perm[i] = -1;
    for (;;) {
        if (depth == nc) {
            deal_state d;
            memset(&d, 0, sizeof(d));
            int ok = 1, trial[32];
            for (int i = 0; i < nc; i++) {
                const int w = perm[i];
                d.items[w] += slot_max[i]; d.nsl[w]++; d.cand[w] = slot_max[i];
                trial[i] = w;
                if (d.items[w] > cap) ok = 0;
            }
            for (int z = nc; ok && z < ns; z++) {
                int pick = -1;
                long pick_cost = 0;
                for (int w = 0; w < LPCN_WAVES; w++) {
                    const int maxsl = LPCN_MAX_SLOTS;
                    if (d.nsl[w] >= maxsl || d.items[w] + slot_max[z] > cap) continue;
                    deal_state t = d;
                    t.items[w] += slot_max[z]; t.nsl[w]++; t.nzr[w]++; t.zr_items[w] += slot_max[z];
                    const long c = deal_wave_cost(dp, &t, w);
                    if (pick < 0 || c < pick_cost) { pick = w; pick_cost = c; }
                }
                if (pick < 0) { ok = 0; break; }
                d.items[pick] += slot_max[z]; d.nsl[pick]++; d.nzr[pick]++; d.zr_items[pick] += slot_max[z];
                trial[z] = pick;
            }
            if (ok) {
                long mx = 0, sum = 0;
                for (int w = 0; w < LPCN_WAVES; w++) { const long c = deal_wave_cost(dp, &d, w); if (c > mx) mx = c; sum += c; }
                if (best_max < 0 || mx < best_max || (mx == best_max && sum < best_sum)) {
                    best_max = mx; best_sum = sum;
                    memcpy(best_wave, trial, sizeof(int) * (size_t)ns);
                }
            }
            depth--;
            if (depth < 0) break;
            used[perm[depth]] = 0;
        }
        /* advance position `depth` to the next free wave */
        int w = perm[depth] + 1;
        while (w < LPCN_WAVES && used[w]) w++;
        if (w >= LPCN_WAVES) {
            perm[depth] = -1;
            depth--;
            if (depth < 0) break;
            used[perm[depth]] = 0;
            continue;
        }
        perm[depth] = w; used[w] = 1;
        depth++;
        if (depth < nc) perm[depth] = -1;
    }
    if (best_max < 0) return -1;
    memcpy(wave_of, best_wave, sizeof(int) * (size_t)ns);
    return 0;
}

/* Deal GRU-A's 144 row groups (8 rows each) to the 8 waves of the sample kernel.
 *  1. sort groups by block count, take them 8 at a time -> 18 "slots" of 64 rows whose lanes
 *     have (nearly) equal trip counts;
 *  2. longest-processing-time assignment of slots to waves (<= 3 slots per wave);
 *  3. per lane and item: the 4 weights of (row, block) and the block's input index. */
static int pack_gru_a(lpcn_model_host *m, int for_fast)
{
    deal_params dpv, *const dp = &dpv;
    enum { NG = LPCN_ROWS_A / 8, NSLOT = NG / 8 };
    row_group g[NG];
    const int *idx = m->a_idx;
    int blk = 0;
    for (int i = 0; i < NG; i++) {
        g[i].group = i; g[i].count = *idx++; g[i].pos = idx; g[i].first_block = blk;
        idx += g[i].count; blk += g[i].count;
    }
    const char *old_i8 = getenv("LPCN_DEAL_I8_OLD");      /* tools: "1" = int8 blobs dealt without early heads (the round-2 scheme) */
    const int deal2 = !m->is_int8 || !(old_i8 && old_i8[0] == '1');      /* split candidate chains (see deal_wave_cost) */
    {
        const char *eh = getenv("LPCN_DEAL_EH");          /* tools: head length of the early candidate items (float default 20: 18 / 20 / 22 / 24 -> 104.4 / 105.0 / 104.1 / 103.3 M samples/s) */
        dp->eh = (eh && *eh) ? atoi(eh) : (m->is_int8 ? LPCN_DEAL_EH_I8 : LPCN_DEAL_EH_F32);     /* int8: round 3, heads on waves 3..7: 6 / 10 / 14 -> 141 / 145 / 147 M samples/s; round 5: lpcnet_engine.h */
        {   /* int8 blobs run two streams per workgroup (two workgroups per CU): only two waves run GRU-B there, every other wave takes a
             * head (a wave that is both -- waves 0, 1 at four streams per workgroup -- runs it behind GRU-B's gate stage; the auto-tune
             * does not pick S = 4 for int8 batches that fill the GPU) */
            const char *hw = getenv("LPCN_DEAL_HW");
            dp->hw = (hw && *hw) ? atoi(hw) : (m->is_int8 ? LPCN_DEAL_HW_I8 : LPCN_WAVES / 2);
            if (dp->hw < 2) dp->hw = 2;
            if (dp->hw > LPCN_WAVES / 2) dp->hw = LPCN_WAVES / 2;
            if (!m->is_int8) dp->hw = LPCN_WAVES / 2;     /* float kernels re-run a head only on waves that never run GRU-B (a gate wave's parked sums would go stale) */
            dp->hmask = (0xFFu << dp->hw) & 0xFFu;
            /* int8 blobs, two streams per workgroup: GRU-B runs on waves LPCN_I8_GBWA / _GBWB (lpcnet_engine.h), every other wave is
             * free in that phase and may carry a head */
            const char *hm = getenv("LPCN_DEAL_HMASK");       /* tools */
            if (m->is_int8 && !for_fast && !(hw && *hw)) dp->hmask = LPCN_DEAL_HMASK_I8;      /* (FAST's own image has no heads and keeps the dealing it was tuned with) */
            if (m->is_int8 && hm && *hm) dp->hmask = (unsigned)strtoul(hm, NULL, 0) & 0xFFu;
        }
        if (for_fast == 1) {                           /* FAST int8 (GRU-B split over all waves: no shadow to hide a head in): 0 / 6 / 14 -> 184 / 163 / 167 M */
            const char *ehf = getenv("LPCN_DEAL_EH_FAST");
            dp->eh = (ehf && *ehf) ? atoi(ehf) : LPCN_DEAL_EH_FAST_I8;
        }
        if (dp->eh < 0) dp->eh = 0;
        if (m->is_int8) { dp->tg = 4200; dp->t0a = 1600; dp->t0b = 900; dp->ci = 250; dp->cu = 220; }
        else            { dp->tg = 5000; dp->t0a = 1600; dp->t0b = 900; dp->ci = 395; dp->cu = 320; }
        dp->tl = 0;
        const char *cm = getenv("LPCN_DEAL_COST");        /* tools: "TG,T0a,T0b,CI,CU[,TL]" of deal_wave_cost */
        if (cm && *cm) sscanf(cm, "%d,%d,%d,%d,%d,%d", &dp->tg, &dp->t0a, &dp->t0b, &dp->ci, &dp->cu, &dp->tl);
    }
    if (deal2) {                                    /* candidate groups first (6 slots), then update/reset groups (12 slots) */
        row_group c[NG], z[NG];
        int nc = 0, nz = 0;
        for (int i = 0; i < NG; i++) { if (g[i].group * 8 >= 2 * LPCN_N_A) c[nc++] = g[i]; else z[nz++] = g[i]; }
        qsort(c, (size_t)nc, sizeof(c[0]), cmp_group_desc);
        qsort(z, (size_t)nz, sizeof(z[0]), cmp_group_desc);
        memcpy(g, c, sizeof(c[0]) * (size_t)nc);
        memcpy(g + nc, z, sizeof(z[0]) * (size_t)nz);
    } else
    qsort(g, NG, sizeof(g[0]), cmp_group_desc);

    /* Slot -> wave assignment with a small cost model of the sample kernel (unit: items).
     * Roles (fp32 blobs; int8 blobs have no "early" waves because their GRU-B is too short to hide a slot):
     *  - "early" waves (ids 4..7): take the largest candidate-only ("all-h") slots.  That slot is computed
     *    one sample ahead in the shadow of GRU-B, so in the gather-dependent part of the sample it is free;
     *  - "first" waves: a remaining all-h slot runs first, while the previous sample's leader work and the
     *    embedding gather of the new sample are in flight (~G items of time): they pay max(slot, G) and
     *    take only one more slot;
     *  - all other waves idle for G until the gather has landed.
     * Every further slot costs its items + ~2 for begin/end work; they are dealt longest-first to the
     * cheapest wave that still has a slot position and register room (items per lane = VGPRs).
     * Finally the waves are renumbered: early waves -> 4..7; of the rest the one with the fewest items
     * becomes wave 0 (it leads the streams), the next wave 1 (it draws the random thresholds). */
    const int G = m->is_int8 ? 30 : 20;
    const int n_early_max = m->is_int8 ? 0 : LPCN_WAVES / 2;      /* measured: 0/2/3/4 early waves -> int8 134/119/120/123, fp32 -/-/96/98 M samples/s */
    int slot_max[NSLOT], slot_allh[NSLOT], wave_of[NSLOT], nslots[LPCN_WAVES] = {0};
    int items[LPCN_WAVES] = {0}, cost[LPCN_WAVES] = {0}, early[LPCN_WAVES] = {0}, maxslots[LPCN_WAVES];
    for (int s = 0; s < NSLOT; s++) {
        slot_max[s] = g[8 * s].count;
        slot_allh[s] = 1;
        wave_of[s] = -1;
        for (int q = 0; q < 8; q++) if (g[8 * s + q].group * 8 < 2 * LPCN_N_A) slot_allh[s] = 0;
    }
    for (int w = 0; w < LPCN_WAVES; w++) { cost[w] = G; maxslots[w] = LPCN_MAX_SLOTS; }
    {
        int n_early = 0;
        for (int s = 0; s < NSLOT; s++) {           /* pass 1: all-h slots onto empty waves (slots are in descending order) */
            if (!slot_allh[s]) continue;
            int best = -1;
            for (int w = 0; w < LPCN_WAVES; w++) if (nslots[w] == 0) { best = w; break; }
            if (best < 0) continue;
            wave_of[s] = best; nslots[best] = 1; items[best] = slot_max[s];
            if (n_early < n_early_max) { early[best] = 1; n_early++; cost[best] = G + 2; }
            else {
                /* a "first" wave keeps two gather register sets in flight while its items run: one more slot only */
                cost[best] = (slot_max[s] > G ? slot_max[s] : G) + 2;
                maxslots[best] = m->is_int8 ? 3 : 2;     /* (the int8 kernel has registers for a third gather set) */
            }
        }
    }
    int cap = 0;                                    /* items per wave = VGPRs: do not exceed the unavoidable maximum */
    for (int s = 0; s < NSLOT; s++) cap += slot_max[s];
    cap = (cap + LPCN_WAVES - 1) / LPCN_WAVES + 2;
    if (cap < slot_max[0]) cap = slot_max[0];
    for (int s = 0; s < NSLOT; s++) {               /* pass 2: longest-processing-time for the rest */
        if (wave_of[s] >= 0) continue;
        int best = -1;
        for (int pass = 0; pass < 2 && best < 0; pass++)
            for (int w = 0; w < LPCN_WAVES; w++)
                if (nslots[w] < maxslots[w] && (pass || items[w] + slot_max[s] <= cap) && (best < 0 || cost[w] < cost[best])) best = w;
        wave_of[s] = best; nslots[best]++; items[best] += slot_max[s]; cost[best] += slot_max[s] + 2;
    }
    {   /* pass 3: renumber */
        int order[LPCN_WAVES], newid[LPCN_WAVES], lo = 0, hi = LPCN_WAVES - 1;
        for (int w = 0; w < LPCN_WAVES; w++) order[w] = w;
        for (int i = 0; i < LPCN_WAVES; i++)       /* early waves last, otherwise ascending item count */
            for (int j = i + 1; j < LPCN_WAVES; j++) {
                const int a = order[i], c = order[j];
                if (early[c] < early[a] || (early[c] == early[a] && items[c] < items[a])) { order[i] = c; order[j] = a; }
            }
        for (int i = 0; i < LPCN_WAVES; i++) { if (early[order[i]]) newid[order[i]] = hi--; else newid[order[i]] = lo++; }
        int items2[LPCN_WAVES];
        for (int w = 0; w < LPCN_WAVES; w++) items2[newid[w]] = items[w];
        for (int w = 0; w < LPCN_WAVES; w++) items[w] = items2[w];
        for (int s = 0; s < NSLOT; s++) wave_of[s] = newid[wave_of[s]];
    }
    if (deal2) {                                    /* float blobs: the enumeration replaces the assignment above */
        int nc = 0, w2[NSLOT];
        while (nc < NSLOT && slot_allh[nc]) nc++;
        static const int caps[] = {30, 32, 36, 40, 48, 64, 80, 96, 0};
        int done = 0;
        for (const int *c = caps; *c && !done; c++) {
            if (slot_max[0] > *c) continue;
            if (deal_v2(dp, slot_max, nc, NSLOT, *c, w2) == 0) done = 1;
        }
        if (done) {
            /* the lightest GRU-B wave leads the streams (wave 0), the next draws the thresholds (wave 1) */
            const int nh = (dp->tl || dp->hmask != ((0xFFu << dp->hw) & 0xFFu)) ? 0 : dp->hw;        /* (waves without a head: interchangeable -- unless the cost model told wave 0 apart) */
            int load4[LPCN_WAVES / 2] = {0}, ord[LPCN_WAVES / 2], newid[LPCN_WAVES];
            for (int sl = 0; sl < NSLOT; sl++) if (w2[sl] < nh) load4[w2[sl]] += slot_max[sl];
            for (int i = 0; i < nh; i++) ord[i] = i;
            for (int i = 0; i < nh; i++)
                for (int j = i + 1; j < nh; j++) if (load4[ord[j]] < load4[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
            for (int w = 0; w < LPCN_WAVES; w++) newid[w] = w;
            for (int i = 0; i < nh; i++) newid[ord[i]] = i;
            for (int w = 0; w < LPCN_WAVES; w++) items[w] = 0;
            for (int sl = 0; sl < NSLOT; sl++) { wave_of[sl] = newid[w2[sl]]; items[wave_of[sl]] += slot_max[sl]; }
        }
    }
    /* (for_fast == 2, the two-group kernel's image: the SAME dealing.  Its start values come from an element-wise pass on waves 4..7, so a row may
     * live on any wave, and a dealing of its own was built -- longest candidate slots on waves 4..7, everything else longest-first to the cheapest
     * wave of a cost model fitted to its phase clocks -- and measured: 135.9 M samples/s against 145.6 M with this one, and tools/deal_search.py
     * --x2 found no neighbour of this one that is faster (round 6; LPCN_DEAL_FORCE_X2 forces a map for such measurements).) */
    {   /* tools (tools/deal_search.py): LPCN_DEAL_FORCE = "w0,w1,...,w17" puts slot i (candidate slots first, both kinds by descending
         * length) on wave wi -- a dealing found by MEASUREMENT can be compared with the cost model's; LPCN_DEAL_PRINT=1 prints the map */
        const char *force = getenv(for_fast == 2 ? "LPCN_DEAL_FORCE_X2" : "LPCN_DEAL_FORCE");
        if (force && *force && deal2 && for_fast != 1) {
            int wf[NSLOT], n = 0, cnt[LPCN_WAVES] = {0}, sum[LPCN_WAVES] = {0}, ok = 1;
            for (const char *q = force; *q && n < NSLOT; n++) { wf[n] = atoi(q); while (*q && *q != ',') q++; if (*q) q++; }
            for (int sl = 0; ok && sl < NSLOT; sl++) {
                if (n != NSLOT || wf[sl] < 0 || wf[sl] >= LPCN_WAVES) { ok = 0; break; }
                cnt[wf[sl]]++; sum[wf[sl]] += slot_max[sl];
                if (cnt[wf[sl]] > LPCN_MAX_SLOTS) ok = 0;
            }
            for (int w = 0; ok && w < LPCN_WAVES; w++) {      /* at most one candidate slot per wave (it runs first / carries the head) */
                int nc = 0;
                for (int sl = 0; sl < NSLOT; sl++) if (wf[sl] == w && slot_allh[sl]) nc++;
                if (nc > 1 || sum[w] > 40) ok = 0;
            }
            if (ok) for (int sl = 0; sl < NSLOT; sl++) wave_of[sl] = wf[sl];
            if (ok) { for (int w = 0; w < LPCN_WAVES; w++) items[w] = sum[w]; }
            else fprintf(stderr, "LPCN_DEAL_FORCE ignored (needs %d waves 0..%d, <= %d slots and one candidate slot per wave, <= 40 items)\n", NSLOT, LPCN_WAVES - 1, LPCN_MAX_SLOTS);
        }
        const char *pr = getenv("LPCN_DEAL_PRINT");
        if (pr && *pr == '1' && for_fast != 1) {
            fprintf(stderr, "LPCN_DEAL slots (length:wave%s%s):", m->is_int8 ? ", int8" : "", for_fast == 2 ? ", two-group kernel" : "");
            for (int sl = 0; sl < NSLOT; sl++) fprintf(stderr, " %s%d:%d", slot_allh[sl] ? "c" : "", slot_max[sl], wave_of[sl]);
            fprintf(stderr, "\n");
        }
    }
    int *load = items;
    int nw = 1;
    for (int w = 0; w < LPCN_WAVES; w++) if (load[w] > nw) nw = load[w];
    m->nw = nw;

    if (m->is_int8) m->pk_a_wq = (int32_t *)calloc((size_t)LPCN_WAVES * nw * 64, sizeof(int32_t));
    else            m->pk_a_w  = (float *)calloc((size_t)LPCN_WAVES * nw * 64 * 4, sizeof(float));
    m->pk_a_blk = (uint8_t *)calloc((size_t)LPCN_WAVES * nw * 64, 1);
    m->pk_a_row = (int32_t *)malloc(sizeof(int32_t) * LPCN_WAVES * LPCN_MAX_SLOTS * 64);
    if ((!m->pk_a_w && !m->pk_a_wq) || !m->pk_a_blk || !m->pk_a_row) return -1;
    for (int i = 0; i < LPCN_WAVES * LPCN_MAX_SLOTS * 64; i++) m->pk_a_row[i] = -1;

    /* Slot order inside a wave.  Float blobs: the candidate-only slot comes first on every wave (it runs while the new
     * sample's gather is in flight); on waves 4..7 the first `head` items of its chains are stored END-ALIGNED in the item
     * array, [nw - head, nw): the kernel runs them one sample ahead in GRU-B's shadow and starts the slot from the parked
     * partial sums.  int8 blobs: no early work (their GRU-B is too short to hide any), slots in dealing order. */
    int slot_at[LPCN_WAVES][LPCN_MAX_SLOTS];
    for (int w = 0; w < LPCN_WAVES; w++) {
        int list[LPCN_MAX_SLOTS], n = 0, hslot = -1;
        for (int k = 0; k < LPCN_MAX_SLOTS; k++) slot_at[w][k] = -1;
        for (int s = 0; s < NSLOT; s++) if (wave_of[s] == w) list[n++] = s;
        if (deal2)
            for (int i = 0; i < n; i++) if (slot_allh[list[i]]) { hslot = list[i]; break; }
        int k = 0;
        if (hslot >= 0) slot_at[w][k++] = hslot;
        for (int i = 0; i < n; i++) if (list[i] != hslot) slot_at[w][k++] = list[i];
        m->pk_a_head[w] = (hslot >= 0) ? deal_head(dp, w, slot_max[hslot]) : 0;
    }
    for (int w = 0; w < LPCN_WAVES; w++) {
        int cur = 0;
        for (int k = 0; k < LPCN_MAX_SLOTS; k++) {
            const int s = slot_at[w][k], j0 = cur;
            const int head = (k == 0) ? m->pk_a_head[w] : 0;       /* chain items [0, head) of slot 0 live at [nw - head, nw) */
            m->pk_a_bound[w][k] = j0;
            m->pk_a_allh[w][k] = 1;
            if (s < 0) continue;
            int allh = 1;
            for (int q = 0; q < 8; q++) {
                const row_group *rg = &g[8 * s + q];
                if (rg->group * 8 < 2 * LPCN_N_A) allh = 0;
                for (int r = 0; r < 8; r++) {
                    int lane = 8 * q + r;
                    m->pk_a_row[(w * LPCN_MAX_SLOTS + k) * 64 + lane] = rg->group * 8 + r;
                    for (int j = 0; j < rg->count; j++) {
                        const int pos = j < head ? nw - head + j : j0 + (j - head);
                        size_t item = ((size_t)w * nw + pos) * 64 + lane;
                        if (m->is_int8) {      /* int8 block = [out 8][in 4] (dump_lpcnet.py:106): the row's 4 bytes are one dword */
                            const signed char *blkq = (const signed char *)m->a_w + (size_t)(rg->first_block + j) * 32;
                            memcpy(&m->pk_a_wq[item], blkq + r * 4, 4);
                        } else {               /* float block = [in 4][out 8] (dump_lpcnet.py:107) */
                            const float *blkw = m->a_w + (size_t)(rg->first_block + j) * 32;
                            for (int c = 0; c < 4; c++) m->pk_a_w[item * 4 + c] = blkw[c * 8 + r];
                        }
                        m->pk_a_blk[item] = (uint8_t)(rg->pos[j] >> 2);
                    }
                }
            }
            m->pk_a_allh[w][k] = allh;
            cur += slot_max[s] - head;
        }
        m->pk_a_bound[w][LPCN_MAX_SLOTS] = cur;
    }
    /* Embedding tables re-ordered to the lane layout: E'[level][slot 0..2][thread 0..511] holds the
     * table entry of the row thread t owns in slot k, so the per-sample gather of one slot is a
     * fully coalesced 256-byte read per wave (2 KB per workgroup) per table and stream. */
    if (for_fast == 2) return 0;                    /* (the two-group kernel reads the tables in the blob's own row order) */
    const float *src[3] = {m->emb_sig, m->emb_pred, m->emb_exc};
    for (int tb = 0; tb < 3; tb++) {
        float *dst = (float *)calloc((size_t)256 * LPCN_MAX_SLOTS * LPCN_WG_THREADS, sizeof(float));
        if (!dst) return -1;
        for (int v = 0; v < 256; v++)
            for (int w = 0; w < LPCN_WAVES; w++)
                for (int lane = 0; lane < 64; lane++)
                    for (int k = 0; k < LPCN_MAX_SLOTS; k++) {
                        int r = m->pk_a_row[(w * LPCN_MAX_SLOTS + k) * 64 + lane];
                        dst[((size_t)v * LPCN_MAX_SLOTS + k) * LPCN_WG_THREADS + w * 64 + lane] = r >= 0 ? src[tb][(size_t)v * LPCN_ROWS_A + r] : 0.f;
                    }
        m->pk_emb[tb] = dst;
    }
    return 0;
}

/* GRU-B input matrix: keep the block-sparse structure, re-block each 8x4 block from the blob's
 * [in 4][out 8] to [out 8][in 4] so that one lane (= one output row) fetches its 4 weights of a
 * block with a single 16-byte LDS read. */
static int pack_gru_b(lpcn_model_host *m)
{
    enum { NG = LPCN_ROWS_B / 8 };
    const size_t cap = (size_t)m->nb_b + 3 * NG + 8;     /* every group padded to a multiple of 4 blocks */
    if (m->is_int8) m->pk_b_wq = (int32_t *)calloc(8 * cap, sizeof(int32_t));
    else            m->pk_b_w = (float *)calloc(32 * cap, sizeof(float));
    m->pk_b_blk = (uint8_t *)calloc(cap, 1);
    m->pk_b_start = (int32_t *)malloc(sizeof(int32_t) * (NG + 1));
    if ((!m->pk_b_w && !m->pk_b_wq) || !m->pk_b_blk || !m->pk_b_start) return -1;
    const int *idx = m->b_idx;
    int src_blk = 0, dst = 0, dense = 1;
    for (int g = 0; g < NG; g++) {
        int cnt = *idx++;
        m->pk_b_start[g] = dst;
        if (cnt != LPCN_N_A / 4) dense = 0;
        for (int j = 0; j < cnt; j++, src_blk++, dst++) {
            if (m->is_int8) {          /* [out 8][in 4] int8: one dword per row; four blocks of a row side by side */
                const signed char *src = (const signed char *)m->b_w + (size_t)src_blk * 32;
                for (int r = 0; r < 8; r++)
                    memcpy(m->pk_b_wq + ((size_t)(dst >> 2) * 8 + r) * 4 + (dst & 3), src + r * 4, 4);
            } else {
                const float *src = m->b_w + (size_t)src_blk * 32;
                float *out = m->pk_b_w + (size_t)dst * 32;
                for (int r = 0; r < 8; r++)
                    for (int c = 0; c < 4; c++) out[r * 4 + c] = src[c * 8 + r];
            }
            if ((*idx >> 2) != j) dense = 0;
            m->pk_b_blk[dst] = (uint8_t)(*idx++ >> 2);
        }
        while (dst & 3) dst++;                           /* zero-weight padding blocks (input block 0) */
    }
    m->pk_b_start[NG] = dst;
    m->nb_b_padded = dst;
    m->b_dense = dense;                                  /* every row group lists all 96 input blocks in order */
    return 0;
}

int lpcn_model_parse(lpcn_model_host *m, const unsigned char *blob, int len)
{
    blob_rec rec[64];
    memset(m, 0, sizeof(*m));
    m->lpc_gamma = 1.0f;
    if (!blob || len <= 0) return -1;
    int n = blob_walk(blob, len, rec, 64);
    if (n <= 0) return -1;

#define NEED_F(field, name, count) \
    do { if (!(m->field = (const float *)blob_need(rec, n, name, sizeof(float) * (size_t)(count)))) return -1; } while (0)
    NEED_F(emb_sig,   "gru_a_embed_sig_weights",  256 * LPCN_ROWS_A);
    NEED_F(emb_pred,  "gru_a_embed_pred_weights", 256 * LPCN_ROWS_A);
    NEED_F(emb_exc,   "gru_a_embed_exc_weights",  256 * LPCN_ROWS_A);
    NEED_F(a_dense_w, "gru_a_dense_feature_weights", LPCN_COND * LPCN_ROWS_A);
    NEED_F(a_dense_b, "gru_a_dense_feature_bias", LPCN_ROWS_A);
    NEED_F(b_dense_w, "gru_b_dense_feature_weights", LPCN_COND * LPCN_ROWS_B);
    NEED_F(b_dense_b, "gru_b_dense_feature_bias", LPCN_ROWS_B);
    NEED_F(conv1_w,   "feature_conv1_weights", 3 * LPCN_FRAME_IN * LPCN_COND);
    NEED_F(conv1_b,   "feature_conv1_bias", LPCN_COND);
    NEED_F(conv2_w,   "feature_conv2_weights", 3 * LPCN_COND * LPCN_COND);
    NEED_F(conv2_b,   "feature_conv2_bias", LPCN_COND);
    NEED_F(pitch_emb, "embed_pitch_weights", 256 * LPCN_PITCH_EMB);
    NEED_F(dense1_w,  "feature_dense1_weights", LPCN_COND * LPCN_COND);
    NEED_F(dense1_b,  "feature_dense1_bias", LPCN_COND);
    NEED_F(dense2_w,  "feature_dense2_weights", LPCN_COND * LPCN_COND);
    NEED_F(dense2_b,  "feature_dense2_bias", LPCN_COND);
    NEED_F(fc_w,      "dual_fc_weights", 256 * 2 * LPCN_N_B);
    NEED_F(fc_b,      "dual_fc_bias", 2 * 256);
    NEED_F(fc_f,      "dual_fc_factor", 2 * 256);
    NEED_F(a_bias,    "sparse_gru_a_bias", 2 * LPCN_ROWS_A);
    NEED_F(a_diag,    "sparse_gru_a_recurrent_weights_diag", LPCN_ROWS_A);
    NEED_F(b_bias,    "gru_b_bias", 2 * LPCN_ROWS_B);
#undef NEED_F
    /* bound by the reference's init although the float path never reads them */
    if (!blob_need(rec, n, "embed_sig_weights", sizeof(float) * 256 * 128)) return -1;
    if (!blob_need(rec, n, "sparse_gru_a_subias", sizeof(float) * 2 * LPCN_ROWS_A)) return -1;
    if (!blob_need(rec, n, "gru_b_subias", sizeof(float) * 2 * LPCN_ROWS_B)) return -1;

    if (!(m->a_idx = blob_need_idx(rec, n, "sparse_gru_a_recurrent_weights_idx", LPCN_N_A, LPCN_ROWS_A, &m->nb_a))) return -1;
    if (!(m->b_idx = blob_need_idx(rec, n, "gru_b_weights_idx", LPCN_N_A, LPCN_ROWS_B, &m->nb_b))) return -1;

    const blob_rec *qa = blob_find(rec, n, "sparse_gru_a_recurrent_weights");
    if (!qa) return -1;
    if ((size_t)qa->size == (size_t)32 * m->nb_a * sizeof(float)) m->is_int8 = 0;
    else if (qa->size == 32 * m->nb_a) m->is_int8 = 1;
    else return -1;
    size_t q = m->is_int8 ? 1 : sizeof(float);
    m->a_w = (const float *)qa->data;
    if (!(m->b_w = (const float *)blob_need(rec, n, "gru_b_weights", q * 32 * (size_t)m->nb_b))) return -1;
    if (!(m->b_rec = (const float *)blob_need(rec, n, "gru_b_recurrent_weights", q * LPCN_ROWS_B * LPCN_N_B))) return -1;

    if (pack_gru_a(m, 0) || pack_gru_b(m)) { lpcn_model_release(m); return -1; }
    return 0;
}

/* The FAST arithmetic's own packing of GRU-A (and of the embedding tables, which follow the row assignment): only where its
 * best dealing differs from PARITY's -- int8 blobs, whose FAST GRU-B leaves no shadow for candidate heads.  `f` becomes a
 * shallow copy of `m` (blob pointers shared) with its own pk_a_* / pk_emb arrays; returns 1 when no separate image is needed,
 * 0 on success, -1 on failure.  Release with lpcn_model_release(f) (the GRU-B packing is not copied). */
int lpcn_model_pack_fast(const lpcn_model_host *m, lpcn_model_host *f)
{
    if (!m->is_int8) return 1;                      /* float blobs: head 0 / 8 / 14 / 20 -> 144.8 / 151.1 / 151.3 / 154.3 M: PARITY's dealing is FAST's best too */
    *f = *m;
    f->pk_a_w = NULL; f->pk_a_wq = NULL; f->pk_a_blk = NULL; f->pk_a_row = NULL;
    f->pk_b_w = NULL; f->pk_b_wq = NULL; f->pk_b_start = NULL; f->pk_b_blk = NULL;
    for (int i = 0; i < 3; i++) f->pk_emb[i] = NULL;
    const int rc = pack_gru_a(f, 1);
    if (rc) { lpcn_model_release(f); return -1; }
    return 0;
}

/* The two-group kernel's own dealing of GRU-A (float blobs; see pack_gru_a).  `f` becomes a shallow copy of `m` with its own pk_a_* arrays
 * (no lane-ordered embedding tables); returns 0, or -1 when the model does not fit (more than LPCN_X2_NW_MAX items on a lane, int8 blob).
 * Release with lpcn_model_release(f). */
int lpcn_model_pack_x2(const lpcn_model_host *m, lpcn_model_host *f)
{
    if (m->is_int8) return -1;
    *f = *m;
    f->pk_a_w = NULL; f->pk_a_wq = NULL; f->pk_a_blk = NULL; f->pk_a_row = NULL;
    f->pk_b_w = NULL; f->pk_b_wq = NULL; f->pk_b_start = NULL; f->pk_b_blk = NULL;
    for (int i = 0; i < 3; i++) f->pk_emb[i] = NULL;
    const int rc = pack_gru_a(f, 2);
    if (rc || f->nw > LPCN_X2_NW_MAX) { lpcn_model_release(f); return -1; }
    return 0;
}

void lpcn_model_release(lpcn_model_host *m)
{
    free(m->pk_a_w); free(m->pk_a_wq); free(m->pk_b_wq); free(m->pk_a_blk); free(m->pk_a_row);
    m->pk_a_wq = NULL; m->pk_b_wq = NULL;
    free(m->pk_b_w); free(m->pk_b_start); free(m->pk_b_blk);
    for (int i = 0; i < 3; i++) { free(m->pk_emb[i]); m->pk_emb[i] = NULL; }
    m->pk_a_w = NULL; m->pk_a_blk = NULL; m->pk_a_row = NULL;
    m->pk_b_w = NULL; m->pk_b_start = NULL; m->pk_b_blk = NULL;
}

/* Consistency check of the device packings against the blob they were built from: re-expands both
 * to dense matrices and compares them bit for bit.  Used by the CPU test-suite and by
 * lpcnet_hip_check_model(); returns 0 when consistent, a positive code naming the first mismatch. */
static int selftest_i8(const lpcn_model_host *m)
{
    int rc = 0;
    signed char *dense = (signed char *)calloc((size_t)LPCN_N_A * LPCN_ROWS_A, 1);
    signed char *packed = (signed char *)calloc((size_t)LPCN_N_A * LPCN_ROWS_A, 1);
    if (!dense || !packed) { rc = 100; goto done; }
    {
        const int *idx = m->a_idx;
        const signed char *w = (const signed char *)m->a_w;
        for (int g = 0; g < LPCN_ROWS_A / 8; g++) {
            int cnt = *idx++;
            for (int j = 0; j < cnt; j++, w += 32) {
                int pos = *idx++;
                for (int r = 0; r < 8; r++)
                    for (int c = 0; c < 4; c++) dense[(size_t)(pos + c) * LPCN_ROWS_A + g * 8 + r] = w[r * 4 + c];
            }
        }
    }
    for (int wv = 0; wv < LPCN_WAVES; wv++)
        for (int k = 0; k < LPCN_MAX_SLOTS; k++)
            for (int lane = 0; lane < 64; lane++) {
                int row = m->pk_a_row[(wv * LPCN_MAX_SLOTS + k) * 64 + lane];
                if (row < 0) continue;
                const int j0 = m->pk_a_bound[wv][k], head = k == 0 ? m->pk_a_head[wv] : 0;     /* slot 0's early items sit end-aligned */
                for (int jj = j0 - head; jj < m->pk_a_bound[wv][k + 1]; jj++) {
                    const int j = jj < j0 ? m->nw + (jj - j0) : jj;
                    size_t item = ((size_t)wv * m->nw + j) * 64 + lane;
                    signed char q[4];
                    memcpy(q, &m->pk_a_wq[item], 4);
                    for (int c = 0; c < 4; c++) if (q[c]) packed[(size_t)(m->pk_a_blk[item] * 4 + c) * LPCN_ROWS_A + row] = q[c];
                }
            }
    if (memcmp(dense, packed, (size_t)LPCN_N_A * LPCN_ROWS_A)) { rc = 5; goto done; }
    memset(dense, 0, (size_t)LPCN_N_A * LPCN_ROWS_B);
    memset(packed, 0, (size_t)LPCN_N_A * LPCN_ROWS_B);
    {
        const int *idx = m->b_idx;
        const signed char *w = (const signed char *)m->b_w;
        for (int g = 0; g < LPCN_ROWS_B / 8; g++) {
            int cnt = *idx++;
            for (int j = 0; j < cnt; j++, w += 32) {
                int pos = *idx++, b = m->pk_b_start[g] + j;
                if (m->pk_b_blk[b] != pos / 4) { rc = 8; goto done; }
                for (int r = 0; r < 8; r++) {
                    signed char q[4];
                    memcpy(q, &m->pk_b_wq[((size_t)(b >> 2) * 8 + r) * 4 + (b & 3)], 4);
                    for (int c = 0; c < 4; c++) {
                        dense[(size_t)(pos + c) * LPCN_ROWS_B + g * 8 + r] = w[r * 4 + c];
                        packed[(size_t)(pos + c) * LPCN_ROWS_B + g * 8 + r] = q[c];
                    }
                }
            }
        }
    }
    if (memcmp(dense, packed, (size_t)LPCN_N_A * LPCN_ROWS_B)) rc = 10;
done:
    free(dense); free(packed);
    return rc;
}

int lpcn_model_selftest(const lpcn_model_host *m)
{
    if (m->is_int8) return selftest_i8(m);
    int rc = 0;
    float *dense = (float *)calloc((size_t)LPCN_N_A * LPCN_ROWS_A, sizeof(float));
    float *packed = (float *)calloc((size_t)LPCN_N_A * LPCN_ROWS_A, sizeof(float));
    unsigned char *seen = (unsigned char *)calloc(LPCN_ROWS_A, 1);
    if (!dense || !packed || !seen) { rc = 100; goto done; }
    {   /* GRU-A from the blob: float blocks [in 4][out 8] */
        const int *idx = m->a_idx;
        const float *w = m->a_w;
        for (int g = 0; g < LPCN_ROWS_A / 8; g++) {
            int cnt = *idx++;
            for (int j = 0; j < cnt; j++, w += 32) {
                int pos = *idx++;
                for (int c = 0; c < 4; c++)
                    for (int r = 0; r < 8; r++) dense[(size_t)(pos + c) * LPCN_ROWS_A + g * 8 + r] = w[c * 8 + r];
            }
        }
    }
    for (int wv = 0; wv < LPCN_WAVES; wv++) {
        for (int k = 0; k < LPCN_MAX_SLOTS; k++) {
            int j0 = m->pk_a_bound[wv][k], j1 = m->pk_a_bound[wv][k + 1];
            if (j0 > j1 || j1 > m->nw) { rc = 1; goto done; }
            for (int lane = 0; lane < 64; lane++) {
                int row = m->pk_a_row[(wv * LPCN_MAX_SLOTS + k) * 64 + lane];
                if (row < 0) continue;
                if (row >= LPCN_ROWS_A || seen[row]) { rc = 2; goto done; }
                seen[row] = 1;
                if (m->pk_a_allh[wv][k] && row < 2 * LPCN_N_A) { rc = 3; goto done; }
                const int head = k == 0 ? m->pk_a_head[wv] : 0;       /* slot 0's early items sit end-aligned */
                if ((wv < (m->pk_emb[0] ? LPCN_WAVES / 2 : LPCN_X2_P0_FIRST) && m->pk_a_head[wv]) || head < 0 || head > LPCN_EARLY_MAX || m->pk_a_bound[wv][LPCN_MAX_SLOTS] + m->pk_a_head[wv] > m->nw) { rc = 7; goto done; }
                for (int jj = j0 - head; jj < j1; jj++) {
                    const int j = jj < j0 ? m->nw + (jj - j0) : jj;
                    size_t item = ((size_t)wv * m->nw + j) * 64 + lane;
                    int p = m->pk_a_blk[item];
                    for (int c = 0; c < 4; c++) {
                        float v = m->pk_a_w[item * 4 + c];
                        if (v != 0.f) packed[(size_t)(p * 4 + c) * LPCN_ROWS_A + row] = v;
                    }
                }
            }
        }
    }
    for (int r = 0; r < LPCN_ROWS_A; r++) if (!seen[r]) { rc = 4; goto done; }
    if (memcmp(dense, packed, sizeof(float) * LPCN_N_A * LPCN_ROWS_A)) { rc = 5; goto done; }
    /* lane-ordered embedding tables (the two-group kernel's image has none) */
    if (m->pk_emb[0]) {
        const float *src[3] = {m->emb_sig, m->emb_pred, m->emb_exc};
        for (int tb = 0; tb < 3; tb++)
            for (int v = 0; v < 256; v += 51)
                for (int t = 0; t < LPCN_WG_THREADS; t++)
                    for (int k = 0; k < LPCN_MAX_SLOTS; k++) {
                        int row = m->pk_a_row[((t >> 6) * LPCN_MAX_SLOTS + k) * 64 + (t & 63)];
                        float e = m->pk_emb[tb][((size_t)v * LPCN_MAX_SLOTS + k) * LPCN_WG_THREADS + t];
                        if (row >= 0 && e != src[tb][(size_t)v * LPCN_ROWS_A + row]) { rc = 6; goto done; }
                    }
    }
    /* GRU-B */
    memset(dense, 0, sizeof(float) * LPCN_N_A * LPCN_ROWS_B);
    memset(packed, 0, sizeof(float) * LPCN_N_A * LPCN_ROWS_B);
    {
        const int *idx = m->b_idx;
        const float *w = m->b_w;
        for (int g = 0; g < LPCN_ROWS_B / 8; g++) {
            int cnt = *idx++;
            if ((m->pk_b_start[g] & 3) || m->pk_b_start[g + 1] - m->pk_b_start[g] < cnt) { rc = 7; goto done; }
            for (int j = 0; j < cnt; j++, w += 32) {
                int pos = *idx++;
                int b = m->pk_b_start[g] + j;
                if (m->pk_b_blk[b] != pos / 4) { rc = 8; goto done; }
                for (int c = 0; c < 4; c++)
                    for (int r = 0; r < 8; r++) {
                        dense[(size_t)(pos + c) * LPCN_ROWS_B + g * 8 + r] = w[c * 8 + r];
                        packed[(size_t)(pos + c) * LPCN_ROWS_B + g * 8 + r] = m->pk_b_w[(size_t)b * 32 + r * 4 + c];
                    }
            }
            for (int b = m->pk_b_start[g] + cnt; b < m->pk_b_start[g + 1]; b++)
                for (int q = 0; q < 32; q++) if (m->pk_b_w[(size_t)b * 32 + q] != 0.f) { rc = 9; goto done; }
        }
    }
    if (memcmp(dense, packed, sizeof(float) * LPCN_N_A * LPCN_ROWS_B)) rc = 10;
done:
    free(dense); free(packed); free(seen);
    return rc;
}
