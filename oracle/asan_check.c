/* oracle/asan_check.c -- TEST INFRASTRUCTURE ONLY: a stand-alone driver that runs the C restatement (knn_oracle.c,
 * grid_oracle.c) under -fsanitize=address,undefined on the inputs of the committed golden vectors (the
 * sanitizer stance for the CPU side; GPU sanitizers are not available on the pool).  tests/test_sanitizers.py writes the
 * inputs of a fixture to a raw file, runs this program, and compares the raw output with the fixture's expected arrays --
 * so the instrumented build is checked for memory errors AND for results in one go.  `make -C oracle asan` builds it.
 *
 * File format (little endian): int64 header[8], then the arrays.
 *   knn :  header = {1, B, n_support, n_queries, K, threads, qpar, 0}; f32 support[B*n1*3], f32 queries[B*n2*3]
 *          output: int64 idx[B*n2*K]   (zero-initialised like knn.pyx:93)
 *   grid:  header = {2, n, fdim, ldim, 0, 0, 0, 0}; f32 dl; f32 points[n*3], f32 features[n*fdim], i32 classes[n*ldim]
 *          output: int64 M; f32 points[M*3], f32 features[M*fdim], i32 classes[M*ldim]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static void* must_read(FILE* f, size_t bytes)
{
    void* p = malloc(bytes ? bytes : 1);
    if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) {
        fprintf(stderr, "asan_check: short read (%zu bytes)\n", bytes);
        exit(3);
    }
    return p;
}

int main(int argc, char** argv)
{
    if (argc != 3) {
        fprintf(stderr, "usage: asan_check <in.bin> <out.bin>\n");
        return 2;
    }
    FILE* in = fopen(argv[1], "rb");
    FILE* out = fopen(argv[2], "wb");
    if (!in || !out) return 2;
    int64_t h[8];
    if (fread(h, sizeof h, 1, in) != 1) return 3;
    if (h[0] == 1) {
        const int64_t B = h[1], n1 = h[2], n2 = h[3], K = h[4];
        float* s = must_read(in, sizeof(float) * (size_t)(B * n1 * 3));
        float* q = must_read(in, sizeof(float) * (size_t)(B * n2 * 3));
        int64_t* idx = calloc((size_t)(B * n2 * K) + 1, sizeof(int64_t));
        if (h[6])
            oracle_knn_batch_qpar(s, q, B, n1, n2, K, idx, (int)h[5]);
        else
            oracle_knn_batch(s, q, B, n1, n2, K, idx, (int)h[5]);
        fwrite(idx, sizeof(int64_t), (size_t)(B * n2 * K), out);
        free(s);
        free(q);
        free(idx);
    } else if (h[0] == 2) {
        const int64_t n = h[1], fdim = h[2], ldim = h[3];
        float dl;
        if (fread(&dl, sizeof dl, 1, in) != 1) return 3;
        float* p = must_read(in, sizeof(float) * (size_t)(n * 3));
        float* f = fdim ? must_read(in, sizeof(float) * (size_t)(n * fdim)) : NULL;
        int32_t* c = ldim ? must_read(in, sizeof(int32_t) * (size_t)(n * ldim)) : NULL;
        const int64_t M = oracle_grid_subsample(p, n, f, fdim, c, ldim, dl, NULL, NULL, NULL);
        float* op = malloc(sizeof(float) * (size_t)(M * 3) + 4);
        float* of = fdim ? malloc(sizeof(float) * (size_t)(M * fdim) + 4) : NULL;
        int32_t* oc = ldim ? malloc(sizeof(int32_t) * (size_t)(M * ldim) + 4) : NULL;
        const int64_t M2 = oracle_grid_subsample(p, n, f, fdim, c, ldim, dl, op, of, oc);
        if (M2 != M) {
            fprintf(stderr, "asan_check: count pass %lld != fill pass %lld\n", (long long)M, (long long)M2);
            return 4;
        }
        fwrite(&M, sizeof M, 1, out);
        fwrite(op, sizeof(float), (size_t)(M * 3), out);
        if (fdim) fwrite(of, sizeof(float), (size_t)(M * fdim), out);
        if (ldim) fwrite(oc, sizeof(int32_t), (size_t)(M * ldim), out);
        free(p); free(f); free(c); free(op); free(of); free(oc);
    } else {
        return 2;
    }
    fclose(in);
    return fclose(out) == 0 ? 0 : 5;
}
