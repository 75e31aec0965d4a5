// asan_host_main.hip -- TEST-ONLY stand-alone driver for the host sanitizer build of the product's host-side kd-tree code
// (kdtree_host.hip + kdtree.h's search routine behind the doors of debug_host.hip): reads a raw K-NN case, runs
// ps_debug_knn_host, writes the indices.  Same file format as oracle/asan_check.c (header {1, B, n1, n2, K, ...}, then support
// and queries as float32) so tests/test_sanitizers.py feeds both programs the same golden inputs.  Compiled host-only
// (`hipcc --cuda-host-only`), never part of libpointseg_hip.so.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "debug_hooks.h"

namespace ps {
static char g_err[512];
void set_error(const char* fmt, ...)  // (the product's lives in context.hip, next to the HIP runtime calls this build leaves out)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace ps

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* in = std::fopen(argv[1], "rb");
    FILE* out = std::fopen(argv[2], "wb");
    if (!in || !out) return 2;
    int64_t h[8];
    if (std::fread(h, sizeof h, 1, in) != 1 || h[0] != 1) return 3;
    const int64_t B = h[1], n1 = h[2], n2 = h[3], K = h[4];
    std::vector<float> s((size_t)(B * n1 * 3)), q((size_t)(B * n2 * 3));
    if (std::fread(s.data(), sizeof(float), s.size(), in) != s.size() || std::fread(q.data(), sizeof(float), q.size(), in) != q.size()) return 3;
    std::vector<int32_t> idx((size_t)(B * n2 * K), 0);
    const int rc = ps_debug_knn_host(s.data(), q.data(), B, n1, n2, K, idx.data());
    if (rc != 0) {
        std::fprintf(stderr, "asan_host_main: %s\n", ps::g_err);
        return 4;
    }
    std::vector<int64_t> wide(idx.begin(), idx.end());
    std::fwrite(wide.data(), sizeof(int64_t), wide.size(), out);
    std::fclose(in);
    return std::fclose(out) == 0 ? 0 : 5;
}
