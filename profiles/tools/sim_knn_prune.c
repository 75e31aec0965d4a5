/* sim_knn_prune.c -- host-side count of what tighter pruning would save the K-NN search (experiment, not product code).
 * Builds the nanoflann-shaped tree with the oracle's builder (oracle/knn_oracle.c, included), then runs every self query in leaf
 * order twice: (A) the reference's pruning rule (far child visited when the accumulated plane bound m2 <= worst), with the product's
 * seeded list; (B) the same walk that ALSO skips any child whose TIGHT bounding box is at squared distance >= worst (no point of it
 * can enter the list: insertion needs d < worst, and worst only decreases).  Reports leaves / inner nodes visited per query, the
 * mean over 64-query groups of the slowest lane's leaf count (what a wave pays), and checks that both walks return identical lists.
 * build: gcc -O2 -I oracle profiles/tools/sim_knn_prune.c -o /tmp/sim/sim -lm ; run: /tmp/sim/sim cloud.f32 n K */
#include "../../oracle/knn_oracle.c"
#include <stdio.h>
#include <math.h>

static float (*box_lo)[3], (*box_hi)[3];
static int mode;
static long n_leaf, n_inner;

static void boxes(const tree_t* t, int32_t node)
{
    const node_t* nd = &t->nodes[node];
    if (nd->axis < 0) {
        for (int a = 0; a < 3; ++a) { box_lo[node][a] = INFINITY; box_hi[node][a] = -INFINITY; }
        for (int32_t i = nd->a; i < nd->b; ++i)
            for (int a = 0; a < 3; ++a) {
                float v = coord(t, t->vind[i], a);
                if (v < box_lo[node][a]) box_lo[node][a] = v;
                if (v > box_hi[node][a]) box_hi[node][a] = v;
            }
        return;
    }
    boxes(t, nd->a);
    boxes(t, nd->b);
    for (int a = 0; a < 3; ++a) {
        box_lo[node][a] = fminf(box_lo[nd->a][a], box_lo[nd->b][a]);
        box_hi[node][a] = fmaxf(box_hi[nd->a][a], box_hi[nd->b][a]);
    }
}

static float box_lb(int32_t node, const float* q)
{
    float s = 0.f;
    for (int a = 0; a < 3; ++a) {
        float d = 0.f;
        if (q[a] < box_lo[node][a]) d = box_lo[node][a] - q[a];
        else if (q[a] > box_hi[node][a]) d = q[a] - box_hi[node][a];
        s += d * d;
    }
    return s * (1.0f - 1e-6f);  /* conservative against rounding */
}

static void search2(const tree_t* t, const float* q, int32_t node, float mindistsq, float* dists, rset_t* r)
{
    const node_t* nd = &t->nodes[node];
    if (mode == 1 && box_lb(node, q) >= r->dist[r->cap - 1]) return;
    if (nd->axis < 0) {
        ++n_leaf;
        for (int32_t i = nd->a; i < nd->b; ++i) {
            int32_t p = t->vind[i];
            float d = 0.f;
            for (int ax = 0; ax < 3; ++ax) {
                float diff = q[ax] - coord(t, p, ax);
                d += diff * diff;
            }
            if (d < r->dist[r->cap - 1]) rset_add(r, d, p);
        }
        return;
    }
    ++n_inner;
    int ax = nd->axis;
    float val = q[ax];
    float diff1 = val - nd->lo, diff2 = val - nd->hi;
    int32_t best, other;
    float cut;
    if (diff1 + diff2 < 0) { best = nd->a; other = nd->b; cut = (val - nd->hi) * (val - nd->hi); }
    else { best = nd->b; other = nd->a; cut = (val - nd->lo) * (val - nd->lo); }
    search2(t, q, best, mindistsq, dists, r);
    float dst = dists[ax];
    mindistsq = mindistsq + cut - dst;
    dists[ax] = cut;
    if (mindistsq <= r->dist[r->cap - 1]) search2(t, q, other, mindistsq, dists, r);
    dists[ax] = dst;
}

int main(int argc, char** argv)
{
    const char* path = argv[1];
    const int64_t n = atol(argv[2]);
    const int K = atoi(argv[3]);
    float* pts = (float*)malloc(sizeof(float) * 3 * n);
    FILE* f = fopen(path, "rb");
    if (!f || fread(pts, sizeof(float) * 3, n, f) != (size_t)n) { fprintf(stderr, "read failed\n"); return 1; }
    fclose(f);
    tree_t t;
    tree_build(&t, pts, n);
    box_lo = malloc(sizeof(float[3]) * t.n_nodes);
    box_hi = malloc(sizeof(float[3]) * t.n_nodes);
    boxes(&t, 0);
    int64_t* res[2];
    for (mode = 0; mode < 2; ++mode) {
        res[mode] = (int64_t*)malloc(sizeof(int64_t) * n * K);
        double sum_leaf = 0, sum_inner = 0, sum_wave_max = 0;
        long wave_max = 0, waves = 0;
        float* dist = malloc(sizeof(float) * K);
        int64_t* idx = malloc(sizeof(int64_t) * K);
        for (int64_t tq = 0; tq < n; ++tq) {
            const float* q = pts + 3 * (int64_t)t.vind[tq];
            /* the product's seed: tightest K-wide window of the 2K-1 leaf-order neighbours */
            float seed = FLT_MAX;
            if (n >= 2 * K - 1) {
                int64_t w0 = tq - (K - 1);
                if (w0 < 0) w0 = 0;
                if (w0 > n - (2 * K - 1)) w0 = n - (2 * K - 1);
                float d[128];
                for (int j = 0; j < 2 * K - 1; ++j) {
                    const float* p = pts + 3 * (int64_t)t.vind[w0 + j];
                    float s = 0.f;
                    for (int a = 0; a < 3; ++a) s += (q[a] - p[a]) * (q[a] - p[a]);
                    d[j] = s;
                }
                float m = INFINITY;
                for (int s0 = 0; s0 < K; ++s0) {
                    float mx = 0.f;
                    for (int j = s0; j < s0 + K; ++j) mx = fmaxf(mx, d[j]);
                    m = fminf(m, mx);
                }
                seed = (m + m * 1e-6f) + 1e-30f;
            }
            rset_t r = {dist, idx, K, K};
            for (int j = 0; j < K; ++j) { dist[j] = seed; idx[j] = 0; }
            float dists[3] = {0.f, 0.f, 0.f};
            n_leaf = 0; n_inner = 0;
            search2(&t, q, 0, 0.f, dists, &r);
            for (int j = 0; j < K; ++j) res[mode][tq * K + j] = idx[j];
            sum_leaf += n_leaf; sum_inner += n_inner;
            if (n_leaf > wave_max) wave_max = n_leaf;
            if ((tq & 63) == 63 || tq == n - 1) { sum_wave_max += wave_max; wave_max = 0; ++waves; }
        }
        printf("mode %c: leaves/query %.2f  inner/query %.2f  slowest-lane leaves per 64-query wave %.2f  (lane efficiency %.2f)\n", mode ? 'B' : 'A',
               sum_leaf / n, sum_inner / n, sum_wave_max / waves, (sum_leaf / n) / (sum_wave_max / waves));
    }
    long diff = 0;
    for (int64_t i = 0; i < n * K; ++i) diff += res[0][i] != res[1][i];
    printf("lists identical: %s (%ld differing slots)\n", diff ? "NO" : "yes", diff);
    return 0;
}
