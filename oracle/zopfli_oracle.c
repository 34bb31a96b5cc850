/* zopfli_oracle.c — TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 *
 * CPU restatement of the Zopfli deflate encoder that deft4j's Zopfli compressors call
 * (deft4j-compress/.../MultiCafeUndZopfliCompressor.java:19-25,33,48-52 — CafeUndZopfli 5cdf283e67, master block
 * 8 MiB, block splitting FIRST / LAST / NONE; MultiJZopfliCompressor.java:18-60,78-85 — jzopfli 0.0.4, five option
 * sets).  Neither dependency is in /root/reference, so this file restates the *published* algorithm (google/zopfli,
 * the version both Java libraries port) from its description: hash chains with the run-length second hash, longest
 * match with per-length distances, greedy lazy LZ77 for the block splitter, the iterated shortest-path "squeeze"
 * with the entropy cost model and the pseudo-random perturbation, boundary package-merge code lengths, the RLE-aware
 * tree optimisation and the stored / fixed / dynamic block choice.
 *
 * Pinning: with log flavour ZOPF_LOG_LIBM and splitting FIRST this file is byte-identical to libzopfli 1.0.3
 * (the in-container proxy SURVEY.md §8c names: /opt/conda/lib/libzopfli.so.1.0.3) on the vectors of
 * tests/golden/zopfli_*.bin, and to the reference's own fixture test/asyoulik/asyoulik-zopfli.txt.gz (5 iterations).
 * Splitting LAST follows the 1.0.0 flow (split the optimal parse afterwards), which 1.0.3 no longer has: unpinned.
 * The Java ports themselves are absent: byte parity with CafeUndZopfli / jzopfli is UNPINNED.
 *
 * Log flavours: ZOPF_LOG_LIBM uses libm's log() as Zopfli does; ZOPF_LOG_PORTABLE uses zopf_portable_log(), a pure
 * IEEE-754 +,-,*,/ routine that the GPU kernels restate operation for operation, so GPU and oracle agree to the bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ZOPF_LOG_LIBM 0
#define ZOPF_LOG_PORTABLE 1
#define ZOPF_SPLIT_FIRST 0
#define ZOPF_SPLIT_LAST 1
#define ZOPF_SPLIT_NONE 2

#define MAXM 258
#define MINM 3
#define WSIZE 32768
#define WMASK 32767
#define NUM_LL 288
#define NUM_D 32
#define LARGE 1e30
#define MAX_CHAIN_HITS 8192

typedef unsigned short u16;

/* ---------------------------------------------------------------- portable log */
/* ln(x) for finite x > 0 using only +,-,*,/ in a fixed order (no FMA: compile with -ffp-contract=off). */
double zopf_portable_log(double x) {
    union { double d; uint64_t u; } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7ff) - 1023;
    v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;   /* m in [1,2) */
    double m = v.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    /* 2*atanh(s) = 2s (1 + z/3 + z^2/5 + ...), |s| <= 0.1716 */
    double p = 1.0 / 27.0;
    p = p * z + 1.0 / 25.0;
    p = p * z + 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}
static int g_logflavor = ZOPF_LOG_LIBM;
static double zlog(double x) { return g_logflavor == ZOPF_LOG_LIBM ? log(x) : zopf_portable_log(x); }

/* ---------------------------------------------------------------- symbol tables (RFC 1951 §3.2.5) */
static int len_symbol(int l) {
    static const u16 base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    int s = 28;
    while (base[s] > l) s--;
    return 257 + s;
}
static int len_extra_bits(int l) {
    static const unsigned char eb[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    return eb[len_symbol(l) - 257];
}
static int len_extra_value(int l) {
    static const u16 base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    return l - base[len_symbol(l) - 257];
}
static int lsym_extra_bits(int s) {
    static const unsigned char eb[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    return eb[s - 257];
}
static int dist_symbol(int d) {
    if (d < 5) return d - 1;
    int l = 31 - __builtin_clz((unsigned)(d - 1));
    int r = ((d - 1) >> (l - 1)) & 1;
    return l * 2 + r;
}
static int dist_extra_bits(int d) {
    if (d < 5) return 0;
    return (31 - __builtin_clz((unsigned)(d - 1))) - 1;
}
static int dist_extra_value(int d) {
    if (d < 5) return 0;
    int l = 31 - __builtin_clz((unsigned)(d - 1));
    return (d - (1 + (1 << l))) & ((1 << (l - 1)) - 1);
}
static int dsym_extra_bits(int s) { return s < 4 ? 0 : s / 2 - 1; }

/* ---------------------------------------------------------------- LZ77 store */
typedef struct {
    u16* litlens;
    u16* dists;
    size_t* pos;
    size_t size, cap;
} Store;
static void store_init(Store* s) { memset(s, 0, sizeof *s); }
static void store_free(Store* s) { free(s->litlens); free(s->dists); free(s->pos); memset(s, 0, sizeof *s); }
static void store_add(Store* s, int litlen, int dist, size_t pos) {
    if (s->size == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 1024;
        s->litlens = (u16*)realloc(s->litlens, s->cap * sizeof(u16));
        s->dists = (u16*)realloc(s->dists, s->cap * sizeof(u16));
        s->pos = (size_t*)realloc(s->pos, s->cap * sizeof(size_t));
    }
    s->litlens[s->size] = (u16)litlen;
    s->dists[s->size] = (u16)dist;
    s->pos[s->size] = pos;
    s->size++;
}
static void store_copy(const Store* a, Store* b) {
    b->size = 0;
    for (size_t i = 0; i < a->size; i++) store_add(b, a->litlens[i], a->dists[i], a->pos[i]);
}
static void store_append(const Store* a, Store* b) {
    for (size_t i = 0; i < a->size; i++) store_add(b, a->litlens[i], a->dists[i], a->pos[i]);
}
static size_t store_byte_range(const Store* s, size_t a, size_t b) {
    if (a == b) return 0;
    size_t l = b - 1;
    return s->pos[l] + (s->dists[l] == 0 ? 1 : s->litlens[l]) - s->pos[a];
}
static void store_histogram(const Store* s, size_t a, size_t b, size_t* ll, size_t* d) {
    memset(ll, 0, NUM_LL * sizeof(size_t));
    memset(d, 0, NUM_D * sizeof(size_t));
    for (size_t i = a; i < b; i++) {
        if (s->dists[i] == 0) ll[s->litlens[i]]++;
        else { ll[len_symbol(s->litlens[i])]++; d[dist_symbol(s->dists[i])]++; }
    }
}

/* ---------------------------------------------------------------- hash chains */
typedef struct {
    int* head; u16* prev; int* hashval; int val;
    int* head2; u16* prev2; int* hashval2; int val2;
    u16* same;
} Hash;
static void hash_alloc(Hash* h) {
    h->head = (int*)malloc(sizeof(int) * 65536);
    h->prev = (u16*)malloc(sizeof(u16) * WSIZE);
    h->hashval = (int*)malloc(sizeof(int) * WSIZE);
    h->head2 = (int*)malloc(sizeof(int) * 65536);
    h->prev2 = (u16*)malloc(sizeof(u16) * WSIZE);
    h->hashval2 = (int*)malloc(sizeof(int) * WSIZE);
    h->same = (u16*)malloc(sizeof(u16) * WSIZE);
}
static void hash_free(Hash* h) {
    free(h->head); free(h->prev); free(h->hashval); free(h->head2); free(h->prev2); free(h->hashval2); free(h->same);
}
static void hash_reset(Hash* h) {
    h->val = 0; h->val2 = 0;
    for (int i = 0; i < 65536; i++) { h->head[i] = -1; h->head2[i] = -1; }
    for (int i = 0; i < WSIZE; i++) { h->prev[i] = (u16)i; h->hashval[i] = -1; h->prev2[i] = (u16)i; h->hashval2[i] = -1; h->same[i] = 0; }
}
static void hash_roll(Hash* h, unsigned char c) { h->val = ((h->val << 5) ^ c) & 32767; }
static void hash_warmup(const unsigned char* a, size_t pos, size_t end, Hash* h) {
    hash_roll(h, a[pos]);
    if (pos + 1 < end) hash_roll(h, a[pos + 1]);
}
static void hash_update(const unsigned char* a, size_t pos, size_t end, Hash* h) {
    u16 hpos = (u16)(pos & WMASK);
    size_t amount = 0;
    hash_roll(h, pos + MINM <= end ? a[pos + MINM - 1] : 0);
    h->hashval[hpos] = h->val;
    if (h->head[h->val] != -1 && h->hashval[h->head[h->val]] == h->val) h->prev[hpos] = (u16)h->head[h->val];
    else h->prev[hpos] = hpos;
    h->head[h->val] = hpos;
    /* run length of the byte at pos */
    if (h->same[(pos - 1) & WMASK] > 1) amount = h->same[(pos - 1) & WMASK] - 1;
    while (pos + amount + 1 < end && a[pos] == a[pos + amount + 1] && amount < 65535) amount++;
    h->same[hpos] = (u16)amount;
    h->val2 = ((h->same[hpos] - MINM) & 255) ^ h->val;
    h->hashval2[hpos] = h->val2;
    if (h->head2[h->val2] != -1 && h->hashval2[h->head2[h->val2]] == h->val2) h->prev2[hpos] = (u16)h->head2[h->val2];
    else h->prev2[hpos] = hpos;
    h->head2[h->val2] = hpos;
}
static void hash_prime(const unsigned char* in, size_t instart, size_t inend, Hash* h) {
    size_t ws = instart > WSIZE ? instart - WSIZE : 0;
    hash_reset(h);
    hash_warmup(in, ws, inend, h);
    for (size_t i = ws; i < instart; i++) hash_update(in, i, inend, h);
}

/* Longest match at pos; sublen[l] = distance chosen for length l (may be NULL).  The match cache of the published
 * code only memoises this function's results, so it is not restated. */
static void find_longest(const Hash* h, const unsigned char* a, size_t pos, size_t size, size_t limit,
                         u16* sublen, u16* distance, u16* length) {
    u16 hpos = (u16)(pos & WMASK), p, pp;
    u16 bestdist = 0, bestlength = 1;
    int chain = MAX_CHAIN_HITS;
    unsigned dist;
    const int* hhead = h->head; const u16* hprev = h->prev; const int* hhashval = h->hashval; int hval = h->val;
    if (size - pos < MINM) { *length = 0; *distance = 0; return; }
    if (pos + limit > size) limit = size - pos;
    pp = (u16)hhead[hval];
    p = hprev[pp];
    dist = p < pp ? pp - p : ((WSIZE - p) + pp);
    while (dist < WSIZE) {
        u16 cur = 0;
        if (dist > 0) {
            const unsigned char* scan = &a[pos];
            const unsigned char* match = &a[pos - dist];
            if (pos + bestlength >= size || scan[bestlength] == match[bestlength]) {
                const unsigned char* e = &a[pos] + limit;
                while (scan != e && *scan == *match) { scan++; match++; }
                cur = (u16)(scan - &a[pos]);
            }
            if (cur > bestlength) {
                if (sublen) for (unsigned j = bestlength + 1; j <= cur; j++) sublen[j] = (u16)dist;
                bestdist = (u16)dist;
                bestlength = cur;
                if (cur >= limit) break;
            }
        }
        if (hhead != h->head2 && bestlength >= h->same[hpos] && h->val2 == h->hashval2[p]) {
            hhead = h->head2; hprev = h->prev2; hhashval = h->hashval2; hval = h->val2;
        }
        pp = p;
        p = hprev[p];
        if (p == pp) break;
        dist += p < pp ? pp - p : ((WSIZE - p) + pp);
        chain--;
        if (chain <= 0) break;
    }
    (void)hhashval; (void)hval;
    *distance = bestdist;
    *length = bestlength;
}

/* ---------------------------------------------------------------- greedy (lazy) LZ77, used by the block splitter and
 * as the squeeze's first statistics */
static int length_score(int length, int distance) { return distance > 1024 ? length - 1 : length; }
static void lz77_greedy(const unsigned char* in, size_t instart, size_t inend, Store* store, Hash* h) {
    u16 leng, dist, dummy[259];
    int lengthscore, prevlengthscore;
    unsigned prev_length = 0, prev_match = 0;
    int match_available = 0;
    if (instart == inend) return;
    hash_prime(in, instart, inend, h);
    for (size_t i = instart; i < inend; i++) {
        hash_update(in, i, inend, h);
        find_longest(h, in, i, inend, MAXM, dummy, &dist, &leng);
        lengthscore = length_score(leng, dist);
        prevlengthscore = length_score((int)prev_length, (int)prev_match);
        if (match_available) {
            match_available = 0;
            if (lengthscore > prevlengthscore + 1) {
                store_add(store, in[i - 1], 0, i - 1);
                if (lengthscore >= MINM && leng < MAXM) {
                    match_available = 1; prev_length = leng; prev_match = dist;
                    continue;
                }
            } else {
                leng = (u16)prev_length; dist = (u16)prev_match;
                store_add(store, leng, dist, i - 1);
                for (unsigned j = 2; j < leng; j++) { i++; hash_update(in, i, inend, h); }
                continue;
            }
        } else if (lengthscore >= MINM && leng < MAXM) {
            match_available = 1; prev_length = leng; prev_match = dist;
            continue;
        }
        if (lengthscore >= MINM) store_add(store, leng, dist, i);
        else { leng = 1; store_add(store, in[i], 0, i); }
        for (unsigned j = 1; j < leng; j++) { i++; hash_update(in, i, inend, h); }
    }
}

/* ---------------------------------------------------------------- length-limited code lengths (boundary package-merge) */
typedef struct PMNode { size_t weight; struct PMNode* tail; int count; } PMNode;
typedef struct { PMNode* next; } PMPool;
static void pm_init_node(size_t w, int c, PMNode* t, PMNode* n) { n->weight = w; n->count = c; n->tail = t; }
static void boundary_pm(PMNode* (*lists)[2], PMNode* leaves, int numsymbols, PMPool* pool, int index) {
    int lastcount = lists[index][1]->count;
    if (index == 0 && lastcount >= numsymbols) return;
    PMNode* newchain = pool->next++;
    PMNode* oldchain = lists[index][1];
    lists[index][0] = oldchain;
    lists[index][1] = newchain;
    if (index == 0) {
        pm_init_node(leaves[lastcount].weight, lastcount + 1, 0, newchain);
    } else {
        size_t sum = lists[index - 1][0]->weight + lists[index - 1][1]->weight;
        if (lastcount < numsymbols && sum > leaves[lastcount].weight) {
            pm_init_node(leaves[lastcount].weight, lastcount + 1, oldchain->tail, newchain);
        } else {
            pm_init_node(sum, lastcount, lists[index - 1][1], newchain);
            boundary_pm(lists, leaves, numsymbols, pool, index - 1);
            boundary_pm(lists, leaves, numsymbols, pool, index - 1);
        }
    }
}
static void boundary_pm_final(PMNode* (*lists)[2], PMNode* leaves, int numsymbols, PMPool* pool, int index) {
    int lastcount = lists[index][1]->count;
    size_t sum = lists[index - 1][0]->weight + lists[index - 1][1]->weight;
    if (lastcount < numsymbols && sum > leaves[lastcount].weight) {
        PMNode* newchain = pool->next;
        PMNode* oldchain = lists[index][1]->tail;
        lists[index][1] = newchain;
        newchain->count = lastcount + 1;
        newchain->tail = oldchain;
    } else {
        lists[index][1]->tail = lists[index - 1][1];
    }
}
static int leaf_cmp(const void* a, const void* b) {
    size_t x = ((const PMNode*)a)->weight, y = ((const PMNode*)b)->weight;
    return x < y ? -1 : (x > y ? 1 : 0);
}
int zopf_length_limited(const size_t* freq, int n, int maxbits, unsigned* bitlengths) {
    PMPool pool;
    int numsymbols = 0;
    PMNode* leaves = (PMNode*)malloc(n * sizeof(PMNode));
    for (int i = 0; i < n; i++) bitlengths[i] = 0;
    for (int i = 0; i < n; i++)
        if (freq[i]) { leaves[numsymbols].weight = freq[i]; leaves[numsymbols].count = i; numsymbols++; }
    if ((1 << maxbits) < numsymbols) { free(leaves); return 1; }
    if (numsymbols == 0) { free(leaves); return 0; }
    if (numsymbols == 1) { bitlengths[leaves[0].count] = 1; free(leaves); return 0; }
    if (numsymbols == 2) { bitlengths[leaves[0].count]++; bitlengths[leaves[1].count]++; free(leaves); return 0; }
    for (int i = 0; i < numsymbols; i++) leaves[i].weight = (leaves[i].weight << 9) | (size_t)leaves[i].count;   /* stable: ties by symbol */
    qsort(leaves, numsymbols, sizeof(PMNode), leaf_cmp);
    for (int i = 0; i < numsymbols; i++) leaves[i].weight >>= 9;
    if (numsymbols - 1 < maxbits) maxbits = numsymbols - 1;
    PMNode* nodes = (PMNode*)malloc(maxbits * 2 * numsymbols * sizeof(PMNode));
    pool.next = nodes;
    PMNode* (*lists)[2] = (PMNode * (*)[2]) malloc(maxbits * sizeof(*lists));
    PMNode* node0 = pool.next++;
    PMNode* node1 = pool.next++;
    pm_init_node(leaves[0].weight, 1, 0, node0);
    pm_init_node(leaves[1].weight, 2, 0, node1);
    for (int i = 0; i < maxbits; i++) { lists[i][0] = node0; lists[i][1] = node1; }
    int runs = 2 * numsymbols - 4;
    for (int i = 0; i < runs - 1; i++) boundary_pm(lists, leaves, numsymbols, &pool, maxbits - 1);
    boundary_pm_final(lists, leaves, numsymbols, &pool, maxbits - 1);
    {   /* extract */
        int counts[16] = {0};
        unsigned end = 16, ptr = 15, value = 1;
        for (PMNode* node = lists[maxbits - 1][1]; node; node = node->tail) counts[--end] = node->count;
        int val = counts[15];
        while (ptr >= end) {
            for (; val > counts[ptr - 1]; val--) bitlengths[leaves[val - 1].count] = value;
            ptr--;
            value++;
        }
    }
    free(lists); free(nodes); free(leaves);
    return 0;
}
static void calc_bit_lengths(const size_t* count, int n, int maxbits, unsigned* out) { zopf_length_limited(count, n, maxbits, out); }
static void lengths_to_symbols(const unsigned* lengths, int n, unsigned maxbits, unsigned* symbols) {
    size_t* bl_count = (size_t*)calloc(maxbits + 1, sizeof(size_t));
    size_t* next_code = (size_t*)calloc(maxbits + 1, sizeof(size_t));
    for (int i = 0; i < n; i++) { symbols[i] = 0; bl_count[lengths[i]]++; }
    size_t code = 0;
    bl_count[0] = 0;
    for (unsigned bits = 1; bits <= maxbits; bits++) { code = (code + bl_count[bits - 1]) << 1; next_code[bits] = code; }
    for (int i = 0; i < n; i++) { unsigned len = lengths[i]; if (len) { symbols[i] = (unsigned)next_code[len]; next_code[len]++; } }
    free(bl_count); free(next_code);
}

/* ---------------------------------------------------------------- bit output */
typedef struct { unsigned char* d; size_t n, cap; unsigned char bp; } Out;
static void out_byte(Out* o, unsigned char b) {
    if (o->n == o->cap) { o->cap = o->cap ? o->cap * 2 : 4096; o->d = (unsigned char*)realloc(o->d, o->cap); }
    o->d[o->n++] = b;
}
static void add_bit(Out* o, int bit) {
    if (o->bp == 0) out_byte(o, 0);
    o->d[o->n - 1] |= (unsigned char)(bit << o->bp);
    o->bp = (o->bp + 1) & 7;
}
static void add_bits(Out* o, unsigned symbol, unsigned length) { for (unsigned i = 0; i < length; i++) add_bit(o, (symbol >> i) & 1); }
static void add_huff(Out* o, unsigned symbol, unsigned length) { for (unsigned i = 0; i < length; i++) add_bit(o, (symbol >> (length - i - 1)) & 1); }

/* ---------------------------------------------------------------- block sizes and dynamic headers */
static void patch_distance_codes(unsigned* d_lengths) {
    int num = 0;
    for (int i = 0; i < 30; i++) { if (d_lengths[i]) num++; if (num >= 2) return; }
    if (num == 0) d_lengths[0] = d_lengths[1] = 1;
    else if (num == 1) d_lengths[d_lengths[0] ? 1 : 0] = 1;
}
/* size of (and optionally emission of) the code-length header for one of the 8 combinations of using 16/17/18 */
static size_t encode_tree(const unsigned* ll_lengths, const unsigned* d_lengths, int use_16, int use_17, int use_18, Out* o) {
    static const unsigned order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    unsigned rle[NUM_LL + NUM_D], rle_bits[NUM_LL + NUM_D];
    size_t rle_size = 0;
    unsigned hlit = 29, hdist = 29, hclen;
    size_t clcounts[19];
    unsigned clcl[19], clsymbols[19];
    for (int i = 0; i < 19; i++) clcounts[i] = 0;
    while (hlit > 0 && ll_lengths[257 + hlit - 1] == 0) hlit--;
    while (hdist > 0 && d_lengths[1 + hdist - 1] == 0) hdist--;
    unsigned hlit2 = hlit + 257;
    unsigned total = hlit2 + hdist + 1;
#define SYM(k) ((k) < hlit2 ? ll_lengths[k] : d_lengths[(k) - hlit2])
    for (unsigned i = 0; i < total; i++) {
        unsigned symbol = SYM(i);
        unsigned count = 1;
        if (use_16 || (symbol == 0 && (use_17 || use_18)))
            for (unsigned j = i + 1; j < total && symbol == SYM(j); j++) count++;
        i += count - 1;
        if (symbol == 0 && count >= 3) {
            if (use_18) while (count >= 11) { unsigned c2 = count > 138 ? 138 : count; rle[rle_size] = 18; rle_bits[rle_size++] = c2 - 11; clcounts[18]++; count -= c2; }
            if (use_17) while (count >= 3) { unsigned c2 = count > 10 ? 10 : count; rle[rle_size] = 17; rle_bits[rle_size++] = c2 - 3; clcounts[17]++; count -= c2; }
        }
        if (use_16 && count >= 4) {
            count--;
            clcounts[symbol]++;
            rle[rle_size] = symbol; rle_bits[rle_size++] = 0;
            while (count >= 3) { unsigned c2 = count > 6 ? 6 : count; rle[rle_size] = 16; rle_bits[rle_size++] = c2 - 3; clcounts[16]++; count -= c2; }
        }
        clcounts[symbol] += count;
        while (count > 0) { rle[rle_size] = symbol; rle_bits[rle_size++] = 0; count--; }
    }
#undef SYM
    calc_bit_lengths(clcounts, 19, 7, clcl);
    hclen = 15;
    while (hclen > 0 && clcounts[order[hclen + 4 - 1]] == 0) hclen--;
    if (o) {
        lengths_to_symbols(clcl, 19, 7, clsymbols);
        add_bits(o, hlit, 5); add_bits(o, hdist, 5); add_bits(o, hclen, 4);
        for (unsigned i = 0; i < hclen + 4; i++) add_bits(o, clcl[order[i]], 3);
        for (size_t i = 0; i < rle_size; i++) {
            add_huff(o, clsymbols[rle[i]], clcl[rle[i]]);
            if (rle[i] == 16) add_bits(o, rle_bits[i], 2);
            else if (rle[i] == 17) add_bits(o, rle_bits[i], 3);
            else if (rle[i] == 18) add_bits(o, rle_bits[i], 7);
        }
    }
    size_t result = 14 + (hclen + 4) * 3;
    for (int i = 0; i < 19; i++) result += clcl[i] * clcounts[i];
    result += clcounts[16] * 2 + clcounts[17] * 3 + clcounts[18] * 7;
    return result;
}
static size_t tree_size(const unsigned* ll, const unsigned* d) {
    size_t best = 0;
    for (int i = 0; i < 8; i++) {
        size_t s = encode_tree(ll, d, i & 1, i & 2, i & 4, 0);
        if (best == 0 || s < best) best = s;
    }
    return best;
}
static void add_dynamic_tree(const unsigned* ll, const unsigned* d, Out* o) {
    int best = 0;
    size_t bestsize = 0;
    for (int i = 0; i < 8; i++) {
        size_t s = encode_tree(ll, d, i & 1, i & 2, i & 4, 0);
        if (bestsize == 0 || s < bestsize) { bestsize = s; best = i; }
    }
    encode_tree(ll, d, best & 1, best & 2, best & 4, o);
}
static size_t symbol_size_counts(const size_t* llc, const size_t* dc, const unsigned* ll, const unsigned* d) {
    size_t r = 0;
    for (int i = 0; i < 256; i++) r += ll[i] * llc[i];
    for (int i = 257; i < 286; i++) r += (ll[i] + lsym_extra_bits(i)) * llc[i];
    for (int i = 0; i < 30; i++) r += (d[i] + dsym_extra_bits(i)) * dc[i];
    r += ll[256];
    return r;
}
static void optimize_for_rle(int length, size_t* counts) {
    int i, k, stride;
    size_t symbol, sum, limit;
    for (; length >= 0; --length) {
        if (length == 0) return;
        if (counts[length - 1] != 0) break;
    }
    int* good = (int*)calloc(length, sizeof(int));
    symbol = counts[0];
    stride = 0;
    for (i = 0; i < length + 1; ++i) {
        if (i == length || counts[i] != symbol) {
            if ((symbol == 0 && stride >= 5) || (symbol != 0 && stride >= 7))
                for (k = 0; k < stride; ++k) good[i - k - 1] = 1;
            stride = 1;
            if (i != length) symbol = counts[i];
        } else ++stride;
    }
    stride = 0;
    limit = counts[0];
    sum = 0;
    for (i = 0; i < length + 1; ++i) {
        size_t ad = 0;
        if (i != length) ad = counts[i] > limit ? counts[i] - limit : limit - counts[i];
        if (i == length || good[i] || ad >= 4) {
            if (stride >= 4 || (stride >= 3 && sum == 0)) {
                int count = (int)((sum + stride / 2) / stride);
                if (count < 1) count = 1;
                if (sum == 0) count = 0;
                for (k = 0; k < stride; ++k) counts[i - k - 1] = count;
            }
            stride = 0;
            sum = 0;
            if (i < length - 3) limit = (counts[i] + counts[i + 1] + counts[i + 2] + counts[i + 3] + 2) / 4;
            else if (i < length) limit = counts[i];
            else limit = 0;
        }
        ++stride;
        if (i != length) sum += counts[i];
    }
    free(good);
}
static double try_optimize_for_rle(const size_t* llc, const size_t* dc, unsigned* ll, unsigned* d) {
    size_t llc2[NUM_LL], dc2[NUM_D];
    unsigned ll2[NUM_LL], d2[NUM_D];
    double treesize = (double)tree_size(ll, d);
    double datasize = (double)symbol_size_counts(llc, dc, ll, d);
    memcpy(llc2, llc, sizeof llc2);
    memcpy(dc2, dc, sizeof dc2);
    optimize_for_rle(NUM_LL, llc2);
    optimize_for_rle(NUM_D, dc2);
    calc_bit_lengths(llc2, NUM_LL, 15, ll2);
    calc_bit_lengths(dc2, NUM_D, 15, d2);
    patch_distance_codes(d2);
    double treesize2 = (double)tree_size(ll2, d2);
    double datasize2 = (double)symbol_size_counts(llc, dc, ll2, d2);
    if (treesize2 + datasize2 < treesize + datasize) {
        memcpy(ll, ll2, sizeof ll2);
        memcpy(d, d2, sizeof d2);
        return treesize2 + datasize2;
    }
    return treesize + datasize;
}
static double dynamic_lengths(const Store* s, size_t a, size_t b, unsigned* ll, unsigned* d) {
    size_t llc[NUM_LL], dc[NUM_D];
    store_histogram(s, a, b, llc, dc);
    llc[256] = 1;
    calc_bit_lengths(llc, NUM_LL, 15, ll);
    calc_bit_lengths(dc, NUM_D, 15, d);
    patch_distance_codes(d);
    return try_optimize_for_rle(llc, dc, ll, d);
}
static void fixed_tree(unsigned* ll, unsigned* d) {
    for (int i = 0; i < 144; i++) ll[i] = 8;
    for (int i = 144; i < 256; i++) ll[i] = 9;
    for (int i = 256; i < 280; i++) ll[i] = 7;
    for (int i = 280; i < 288; i++) ll[i] = 8;
    for (int i = 0; i < 32; i++) d[i] = 5;
}
static double block_size(const Store* s, size_t a, size_t b, int btype) {
    unsigned ll[NUM_LL], d[NUM_D];
    double result = 3;
    if (btype == 0) {
        size_t length = store_byte_range(s, a, b);
        size_t rem = length % 65535;
        size_t blocks = length / 65535 + (rem ? 1 : 0);
        return (double)(blocks * 5 * 8 + length * 8);
    }
    if (btype == 1) {
        size_t llc[NUM_LL], dc[NUM_D];
        fixed_tree(ll, d);
        store_histogram(s, a, b, llc, dc);
        result += (double)symbol_size_counts(llc, dc, ll, d);
    } else {
        result += dynamic_lengths(s, a, b, ll, d);
    }
    return result;
}
static double block_size_auto(const Store* s, size_t a, size_t b) {
    double unc = block_size(s, a, b, 0);
    double fixedc = s->size > 1000 ? unc : block_size(s, a, b, 1);
    double dyn = block_size(s, a, b, 2);
    return (unc < fixedc && unc < dyn) ? unc : (fixedc < dyn ? fixedc : dyn);
}

/* ---------------------------------------------------------------- block splitting */
typedef struct { const Store* s; size_t start, end; } SplitCtx;
static double split_cost(size_t i, const SplitCtx* c) { return block_size_auto(c->s, c->start, i) + block_size_auto(c->s, i, c->end); }
static size_t find_minimum(const SplitCtx* c, size_t start, size_t end, double* smallest) {
    if (end - start < 1024) {
        double best = LARGE;
        size_t result = start;
        for (size_t i = start; i < end; i++) { double v = split_cost(i, c); if (v < best) { best = v; result = i; } }
        *smallest = best;
        return result;
    }
    enum { NUM = 9 };
    size_t p[NUM];
    double vp[NUM];
    double lastbest = LARGE;
    size_t pos = start;
    for (;;) {
        if (end - start <= NUM) break;
        for (int i = 0; i < NUM; i++) { p[i] = start + (i + 1) * ((end - start) / (NUM + 1)); vp[i] = split_cost(p[i], c); }
        int besti = 0;
        double best = vp[0];
        for (int i = 1; i < NUM; i++) if (vp[i] < best) { best = vp[i]; besti = i; }
        if (best > lastbest) break;
        start = besti == 0 ? start : p[besti - 1];
        end = besti == NUM - 1 ? end : p[besti + 1];
        pos = p[besti];
        lastbest = best;
    }
    *smallest = lastbest;
    return pos;
}
static void add_sorted(size_t value, size_t** out, size_t* n) {
    *out = (size_t*)realloc(*out, (*n + 1) * sizeof(size_t));
    (*out)[*n] = value;
    (*n)++;
    for (size_t i = 0; i + 1 < *n; i++)
        if ((*out)[i] > value) {
            for (size_t j = *n - 1; j > i; j--) (*out)[j] = (*out)[j - 1];
            (*out)[i] = value;
            break;
        }
}
static int find_largest_splittable(size_t lzsize, const unsigned char* done, const size_t* sp, size_t np, size_t* lstart, size_t* lend) {
    size_t longest = 0;
    int found = 0;
    for (size_t i = 0; i <= np; i++) {
        size_t start = i == 0 ? 0 : sp[i - 1];
        size_t end = i == np ? lzsize - 1 : sp[i];
        if (!done[start] && end - start > longest) { *lstart = start; *lend = end; found = 1; longest = end - start; }
    }
    return found;
}
static void block_split_lz77(const Store* s, size_t maxblocks, size_t** sp, size_t* np) {
    if (s->size < 10) return;
    unsigned char* done = (unsigned char*)calloc(s->size, 1);
    size_t lstart = 0, lend = s->size, numblocks = 1;
    for (;;) {
        SplitCtx c;
        double splitcost, origcost;
        if (maxblocks > 0 && numblocks >= maxblocks) break;
        c.s = s; c.start = lstart; c.end = lend;
        size_t llpos = find_minimum(&c, lstart + 1, lend, &splitcost);
        origcost = block_size_auto(s, lstart, lend);
        if (splitcost > origcost || llpos == lstart + 1 || llpos == lend) done[lstart] = 1;
        else { add_sorted(llpos, sp, np); numblocks++; }
        if (!find_largest_splittable(s->size, done, *sp, *np, &lstart, &lend)) break;
        if (lend - lstart < 10) break;
    }
    free(done);
}
static void block_split(const unsigned char* in, size_t instart, size_t inend, size_t maxblocks, size_t** sp, size_t* np) {
    Store store;
    Hash h;
    size_t* lzp = 0;
    size_t nlz = 0;
    store_init(&store);
    hash_alloc(&h);
    *np = 0; *sp = 0;
    lz77_greedy(in, instart, inend, &store, &h);
    block_split_lz77(&store, maxblocks, &lzp, &nlz);
    size_t pos = instart;
    if (nlz > 0) {
        for (size_t i = 0; i < store.size; i++) {
            size_t length = store.dists[i] == 0 ? 1 : store.litlens[i];
            if (lzp[*np] == i) {
                *sp = (size_t*)realloc(*sp, (*np + 1) * sizeof(size_t));
                (*sp)[(*np)++] = pos;
                if (*np == nlz) break;
            }
            pos += length;
        }
    }
    free(lzp);
    store_free(&store);
    hash_free(&h);
}

/* ---------------------------------------------------------------- squeeze */
typedef struct {
    size_t litlens[NUM_LL], dists[NUM_D];
    double ll_symbols[NUM_LL], d_symbols[NUM_D];
} Stats;
static void calc_entropy(const size_t* count, int n, double* bitlengths) {
    static const double kInvLog2 = 1.4426950408889;
    unsigned sum = 0;
    for (int i = 0; i < n; ++i) sum += (unsigned)count[i];
    double log2sum = (sum == 0 ? zlog((double)n) : zlog((double)sum)) * kInvLog2;
    for (int i = 0; i < n; ++i) {
        if (count[i] == 0) bitlengths[i] = log2sum;
        else bitlengths[i] = log2sum - zlog((double)count[i]) * kInvLog2;
        if (bitlengths[i] < 0 && bitlengths[i] > -1e-5) bitlengths[i] = 0;
    }
}
static void calc_statistics(Stats* st) { calc_entropy(st->litlens, NUM_LL, st->ll_symbols); calc_entropy(st->dists, NUM_D, st->d_symbols); }
static void get_statistics(const Store* s, Stats* st) {
    for (size_t i = 0; i < s->size; i++) {
        if (s->dists[i] == 0) st->litlens[s->litlens[i]]++;
        else { st->litlens[len_symbol(s->litlens[i])]++; st->dists[dist_symbol(s->dists[i])]++; }
    }
    st->litlens[256] = 1;
    calc_statistics(st);
}
typedef struct { unsigned m_w, m_z; } Ran;
static unsigned ran(Ran* r) {
    r->m_z = 36969 * (r->m_z & 65535) + (r->m_z >> 16);
    r->m_w = 18000 * (r->m_w & 65535) + (r->m_w >> 16);
    return (r->m_z << 16) + r->m_w;
}
static void randomize_freqs(Ran* r, size_t* freqs, int n) {
    for (int i = 0; i < n; i++)
        if ((ran(r) >> 4) % 3 == 0) freqs[i] = freqs[ran(r) % n];
}
typedef double (*CostFn)(unsigned litlen, unsigned dist, const void* ctx);
static double cost_fixed(unsigned litlen, unsigned dist, const void* unused) {
    (void)unused;
    if (dist == 0) return litlen <= 143 ? 8 : 9;
    int dbits = dist_extra_bits((int)dist), lbits = len_extra_bits((int)litlen), lsym = len_symbol((int)litlen);
    int cost = 0;
    if (lsym <= 279) cost += 7; else cost += 8;
    cost += 5;
    return cost + dbits + lbits;
}
static double cost_stat(unsigned litlen, unsigned dist, const void* ctx) {
    const Stats* st = (const Stats*)ctx;
    if (dist == 0) return st->ll_symbols[litlen];
    int lsym = len_symbol((int)litlen), lbits = len_extra_bits((int)litlen);
    int dsym = dist_symbol((int)dist), dbits = dist_extra_bits((int)dist);
    return lbits + dbits + st->ll_symbols[lsym] + st->d_symbols[dsym];
}
static double model_min_cost(CostFn f, const void* ctx) {
    static const int dsymbols[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    double mincost = LARGE;
    int bestlength = 0, bestdist = 0;
    for (int i = 3; i < 259; i++) { double c = f(i, 1, ctx); if (c < mincost) { bestlength = i; mincost = c; } }
    mincost = LARGE;
    for (int i = 0; i < 30; i++) { double c = f(3, dsymbols[i], ctx); if (c < mincost) { bestdist = dsymbols[i]; mincost = c; } }
    return f(bestlength, bestdist, ctx);
}
static double best_lengths(const unsigned char* in, size_t instart, size_t inend, CostFn f, const void* ctx,
                           u16* length_array, Hash* h, float* costs) {
    size_t blocksize = inend - instart;
    u16 leng, dist, sublen[259];
    double mincost = model_min_cost(f, ctx);
    if (instart == inend) return 0;
    hash_prime(in, instart, inend, h);
    for (size_t i = 1; i < blocksize + 1; i++) costs[i] = (float)LARGE;
    costs[0] = 0;
    length_array[0] = 0;
    for (size_t i = instart; i < inend; i++) {
        size_t j = i - instart;
        hash_update(in, i, inend, h);
        /* inside a long run of one byte value: take 258-byte matches at distance 1 without searching */
        if (h->same[i & WMASK] > MAXM * 2 && i > instart + MAXM + 1 && i + MAXM * 2 + 1 < inend && h->same[(i - MAXM) & WMASK] > MAXM) {
            double symbolcost = f(MAXM, 1, ctx);
            for (int k = 0; k < MAXM; k++) {
                costs[j + MAXM] = (float)(costs[j] + symbolcost);
                length_array[j + MAXM] = MAXM;
                i++; j++;
                hash_update(in, i, inend, h);
            }
        }
        find_longest(h, in, i, inend, MAXM, sublen, &dist, &leng);
        if (i + 1 <= inend) {
            double nc = f(in[i], 0, ctx) + costs[j];
            if (nc < costs[j + 1]) { costs[j + 1] = (float)nc; length_array[j + 1] = 1; }
        }
        size_t kend = leng < inend - i ? leng : inend - i;
        double mincostaddcostj = mincost + costs[j];
        for (size_t k = 3; k <= kend; k++) {
            if (costs[j + k] <= mincostaddcostj) continue;
            double nc = f((unsigned)k, sublen[k], ctx) + costs[j];
            if (nc < costs[j + k]) { costs[j + k] = (float)nc; length_array[j + k] = (u16)k; }
        }
    }
    return costs[blocksize];
}
static void optimal_run(const unsigned char* in, size_t instart, size_t inend, u16* length_array, CostFn f, const void* ctx,
                        Store* store, Hash* h, float* costs) {
    best_lengths(in, instart, inend, f, ctx, length_array, h, costs);
    size_t size = inend - instart;
    if (size == 0) return;
    /* trace the chosen lengths backwards, then follow them forwards looking each distance up again */
    size_t npath = 0;
    u16* path = (u16*)malloc((size + 1) * sizeof(u16));
    for (size_t idx = size; idx > 0; idx -= length_array[idx]) path[npath++] = length_array[idx];
    hash_prime(in, instart, inend, h);
    size_t pos = instart;
    for (size_t k = npath; k-- > 0;) {
        u16 length = path[k], dummy, dist;
        hash_update(in, pos, inend, h);
        if (length >= MINM) {
            find_longest(h, in, pos, inend, length, 0, &dist, &dummy);
            store_add(store, length, dist, pos);
        } else {
            length = 1;
            store_add(store, in[pos], 0, pos);
        }
        for (unsigned j = 1; j < length; j++) hash_update(in, pos + j, inend, h);
        pos += length;
    }
    free(path);
}
static void lz77_optimal(const unsigned char* in, size_t instart, size_t inend, int numiterations, Store* store) {
    size_t blocksize = inend - instart;
    u16* length_array = (u16*)malloc(sizeof(u16) * (blocksize + 1));
    float* costs = (float*)malloc(sizeof(float) * (blocksize + 1));
    Store cur;
    Hash h;
    Stats stats, beststats, laststats;
    double cost, bestcost = LARGE, lastcost = 0;
    Ran rs = {1, 2};
    int lastrandomstep = -1;
    memset(&stats, 0, sizeof stats);
    memset(&beststats, 0, sizeof beststats);
    store_init(&cur);
    hash_alloc(&h);
    lz77_greedy(in, instart, inend, &cur, &h);
    get_statistics(&cur, &stats);
    for (int i = 0; i < numiterations; i++) {
        cur.size = 0;
        optimal_run(in, instart, inend, length_array, cost_stat, &stats, &cur, &h, costs);
        cost = block_size(&cur, 0, cur.size, 2);
        if (cost < bestcost) { store_copy(&cur, store); beststats = stats; bestcost = cost; }
        laststats = stats;
        memset(stats.litlens, 0, sizeof stats.litlens);
        memset(stats.dists, 0, sizeof stats.dists);
        get_statistics(&cur, &stats);
        if (lastrandomstep != -1) {
            for (int k = 0; k < NUM_LL; k++) stats.litlens[k] = (size_t)(stats.litlens[k] * 1.0 + laststats.litlens[k] * 0.5);
            for (int k = 0; k < NUM_D; k++) stats.dists[k] = (size_t)(stats.dists[k] * 1.0 + laststats.dists[k] * 0.5);
            stats.litlens[256] = 1;
            calc_statistics(&stats);
        }
        if (i > 5 && cost == lastcost) {
            stats = beststats;
            randomize_freqs(&rs, stats.litlens, NUM_LL);
            randomize_freqs(&rs, stats.dists, NUM_D);
            stats.litlens[256] = 1;
            calc_statistics(&stats);
            lastrandomstep = i;
        }
        lastcost = cost;
    }
    free(length_array); free(costs);
    store_free(&cur);
    hash_free(&h);
}
static void lz77_optimal_fixed(const unsigned char* in, size_t instart, size_t inend, Store* store) {
    size_t blocksize = inend - instart;
    u16* length_array = (u16*)malloc(sizeof(u16) * (blocksize + 1));
    float* costs = (float*)malloc(sizeof(float) * (blocksize + 1));
    Hash h;
    hash_alloc(&h);
    optimal_run(in, instart, inend, length_array, cost_fixed, 0, store, &h, costs);
    free(length_array); free(costs);
    hash_free(&h);
}

/* ---------------------------------------------------------------- block emission */
static void add_lz77_data(const Store* s, size_t a, size_t b, const unsigned* llsym, const unsigned* ll, const unsigned* dsym, const unsigned* d, Out* o) {
    for (size_t i = a; i < b; i++) {
        unsigned dist = s->dists[i], litlen = s->litlens[i];
        if (dist == 0) add_huff(o, llsym[litlen], ll[litlen]);
        else {
            unsigned lls = (unsigned)len_symbol((int)litlen), ds = (unsigned)dist_symbol((int)dist);
            add_huff(o, llsym[lls], ll[lls]);
            add_bits(o, (unsigned)len_extra_value((int)litlen), (unsigned)len_extra_bits((int)litlen));
            add_huff(o, dsym[ds], d[ds]);
            add_bits(o, (unsigned)dist_extra_value((int)dist), (unsigned)dist_extra_bits((int)dist));
        }
    }
}
static void add_stored(int final, const unsigned char* in, size_t instart, size_t inend, Out* o) {
    size_t pos = instart;
    for (;;) {
        unsigned short blocksize = 65535;
        if (pos + blocksize > inend) blocksize = (unsigned short)(inend - pos);
        int currentfinal = pos + blocksize >= inend;
        unsigned short nlen = (unsigned short)~blocksize;
        add_bit(o, final && currentfinal);
        add_bit(o, 0); add_bit(o, 0);
        o->bp = 0;
        out_byte(o, blocksize % 256); out_byte(o, (blocksize / 256) % 256);
        out_byte(o, nlen % 256); out_byte(o, (nlen / 256) % 256);
        for (size_t i = 0; i < blocksize; i++) out_byte(o, in[pos + i]);
        if (currentfinal) break;
        pos += blocksize;
    }
}
static void add_block(const unsigned char* in, int btype, int final, const Store* s, size_t a, size_t b, Out* o) {
    unsigned ll[NUM_LL], d[NUM_D], llsym[NUM_LL], dsym[NUM_D];
    if (btype == 0) {
        size_t length = store_byte_range(s, a, b);
        size_t pos = a == b ? 0 : s->pos[a];
        add_stored(final, in, pos, pos + length, o);
        return;
    }
    add_bit(o, final); add_bit(o, btype & 1); add_bit(o, (btype & 2) >> 1);
    if (btype == 1) fixed_tree(ll, d);
    else { dynamic_lengths(s, a, b, ll, d); add_dynamic_tree(ll, d, o); }
    lengths_to_symbols(ll, NUM_LL, 15, llsym);
    lengths_to_symbols(d, NUM_D, 15, dsym);
    add_lz77_data(s, a, b, llsym, ll, dsym, d, o);
    add_huff(o, llsym[256], ll[256]);
}
static void add_block_auto(const unsigned char* in, int final, const Store* s, size_t a, size_t b, Out* o) {
    double unc = block_size(s, a, b, 0), fixedc = block_size(s, a, b, 1), dyn = block_size(s, a, b, 2);
    int expensivefixed = (s->size < 1000) || fixedc <= dyn * 1.1;
    Store fs;
    if (a == b) { add_bits(o, (unsigned)final, 1); add_bits(o, 1, 2); add_bits(o, 0, 7); return; }
    store_init(&fs);
    if (expensivefixed) {
        size_t instart = s->pos[a], inend = instart + store_byte_range(s, a, b);
        lz77_optimal_fixed(in, instart, inend, &fs);
        fixedc = block_size(&fs, 0, fs.size, 1);
    }
    if (unc < fixedc && unc < dyn) add_block(in, 0, final, s, a, b, o);
    else if (fixedc < dyn) { if (expensivefixed) add_block(in, 1, final, &fs, 0, fs.size, o); else add_block(in, 1, final, s, a, b, o); }
    else add_block(in, 2, final, s, a, b, o);
    store_free(&fs);
}

/* ---------------------------------------------------------------- one master block */
static void deflate_part(const unsigned char* in, size_t instart, size_t inend, int final, int iterations, int splitting,
                         size_t maxblocks, Out* o) {
    size_t *spu = 0, *sp = 0, npoints = 0;
    Store lz;
    store_init(&lz);
    if (splitting == ZOPF_SPLIT_FIRST) {
        double totalcost = 0;
        block_split(in, instart, inend, maxblocks, &spu, &npoints);
        sp = (size_t*)malloc((npoints + 1) * sizeof(size_t));
        for (size_t i = 0; i <= npoints; i++) {
            size_t start = i == 0 ? instart : spu[i - 1], end = i == npoints ? inend : spu[i];
            Store st;
            store_init(&st);
            lz77_optimal(in, start, end, iterations, &st);
            totalcost += block_size_auto(&st, 0, st.size);
            store_append(&st, &lz);
            if (i < npoints) sp[i] = lz.size;
            store_free(&st);
        }
        if (npoints > 1) {   /* second attempt: split the final parse, keep it if cheaper */
            size_t *sp2 = 0, np2 = 0;
            double totalcost2 = 0;
            block_split_lz77(&lz, maxblocks, &sp2, &np2);
            for (size_t i = 0; i <= np2; i++) {
                size_t start = i == 0 ? 0 : sp2[i - 1], end = i == np2 ? lz.size : sp2[i];
                totalcost2 += block_size_auto(&lz, start, end);
            }
            if (totalcost2 < totalcost) { free(sp); sp = sp2; npoints = np2; }
            else free(sp2);
        }
    } else {
        lz77_optimal(in, instart, inend, iterations, &lz);
        if (splitting == ZOPF_SPLIT_LAST) block_split_lz77(&lz, maxblocks, &sp, &npoints);
    }
    for (size_t i = 0; i <= npoints; i++) {
        size_t start = i == 0 ? 0 : sp[i - 1], end = i == npoints ? lz.size : sp[i];
        add_block_auto(in, i == npoints && final, &lz, start, end, o);
    }
    free(sp); free(spu);
    store_free(&lz);
}

/* Raw deflate stream of in[0..n).  master = master block size (1000000 in libzopfli / jzopfli, 8 << 20 in deft4j's
 * CafeUndZopfli call).  Returns 0 and a malloc'd buffer (zopf_free). */
int zopf_deflate(const unsigned char* in, size_t n, int iterations, int splitting, int maxblocks, size_t master, int logflavor,
                 unsigned char** out, size_t* outlen) {
    Out o;
    memset(&o, 0, sizeof o);
    g_logflavor = logflavor;
    if (master == 0) master = n ? n : 1;
    size_t i = 0;
    do {
        int masterfinal = i + master >= n;
        size_t size = masterfinal ? n - i : master;
        deflate_part(in, i, i + size, masterfinal, iterations, splitting, (size_t)maxblocks, &o);
        i += size;
    } while (i < n);
    *out = o.d;
    *outlen = o.n;
    return 0;
}
void zopf_free(void* p) { free(p); }

/* Pieces exposed for the tests that pin them one by one. */
void zopf_optimize_for_rle(int length, size_t* counts) { optimize_for_rle(length, counts); }
size_t zopf_greedy(const unsigned char* in, size_t instart, size_t inend, u16* litlens, u16* dists, size_t cap) {
    Store s;
    Hash h;
    store_init(&s);
    hash_alloc(&h);
    lz77_greedy(in, instart, inend, &s, &h);
    size_t n = s.size;
    for (size_t i = 0; i < n && i < cap; i++) { litlens[i] = s.litlens[i]; dists[i] = s.dists[i]; }
    store_free(&s);
    hash_free(&h);
    return n;
}
/* per-position longest match + per-length distances: out_len[i], out_dist[i], sublen[i*259 + l] (l = 3..len) */
void zopf_match_table(const unsigned char* in, size_t instart, size_t inend, u16* out_len, u16* out_dist, u16* sublen) {
    Hash h;
    hash_alloc(&h);
    hash_prime(in, instart, inend, &h);
    for (size_t i = instart; i < inend; i++) {
        u16 sl[259];
        memset(sl, 0, sizeof sl);
        hash_update(in, i, inend, &h);
        find_longest(&h, in, i, inend, MAXM, sl, &out_dist[i - instart], &out_len[i - instart]);
        if (sublen) memcpy(&sublen[(i - instart) * 259], sl, sizeof sl);
    }
    hash_free(&h);
}
