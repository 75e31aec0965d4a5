// common.h -- context, workspace arena, error plumbing and per-stage timing shared by every .hip file of
// libpointseg_hip.so.  Nothing here is visible through the C ABI (include/pointseg.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pointseg.h"

namespace ps {

void set_error(const char* fmt, ...);

#define PS_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            ps::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PS_EHIP;                                                                           \
        }                                                                                             \
    } while (0)

#define PS_CHECK(cond, ...)             \
    do {                                \
        if (!(cond)) {                  \
            ps::set_error(__VA_ARGS__); \
            return PS_EINVAL;           \
        }                               \
    } while (0)

#define PS_TRY(expr)                \
    do {                            \
        int _rc = (expr);           \
        if (_rc != PS_OK) return _rc; \
    } while (0)

// A grow-only device buffer: sized to the high-water mark, then reused (no hipMalloc on the steady-state path).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};

// Bump allocator over one DevBuf.  Usage per API call: plan sizes with `dry` = true, reserve, then carve.
struct Arena {
    DevBuf buf;
    size_t off = 0;
    bool dry = false;
    void begin(bool dry_run) { off = 0; dry = dry_run; }
    template <class T>
    T* take(size_t count)
    {
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        T* r = dry ? nullptr : reinterpret_cast<T*>(static_cast<char*>(buf.p) + off);
        off += bytes;
        return r;
    }
};

struct StageTimer {
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    int64_t launches = 0;
};

}  // namespace ps

struct ps_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    ps::Arena knn_arena;     // kd-trees + query scratch
    ps::Arena net_arena;     // activations of the forward pass
    ps::DevBuf stage_in;     // host<->device staging for device_ptrs == 0 calls
    ps::DevBuf stage_out;
    ps::DevBuf ops_ws;       // packed weights of ps_op_conv1x1
    // timing
    bool timing = false;
    std::vector<ps::StageTimer> stages;
    std::vector<hipEvent_t> event_pool;
    size_t event_next = 0;
    int cur_stage = -1;
    hipEvent_t cur_start = nullptr;

    hipEvent_t get_event();
    void stage_begin(const char* name);
    void stage_end(int launches);
};

namespace ps {
// RAII span: times everything enqueued on ctx->stream between construction and destruction when timing is armed.
struct Stage {
    ps_context* c;
    int n;
    Stage(ps_context* ctx, const char* name, int launches = 1) : c(ctx), n(launches)
    {
        if (c->timing) c->stage_begin(name);
    }
    ~Stage()
    {
        if (c->timing) c->stage_end(n);
    }
};

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
}  // namespace ps
