// context.hip -- ps_context lifetime, error text, workspace buffers, hipEvent stage timing.
#include "common.h"

namespace ps {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return PS_OK;
    if (p) {
        // the stream may still be reading the old block
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            set_error("hipDeviceSynchronize failed: %s", hipGetErrorString(e));
            return PS_EHIP;
        }
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes + bytes / 8 + (1u << 20);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("workspace hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return PS_ENOMEM;
    }
    cap = want;
    return PS_OK;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

}  // namespace ps

hipEvent_t ps_context::get_event()
{
    if (event_next == event_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        event_pool.push_back(e);
    }
    return event_pool[event_next++];
}

void ps_context::stage_begin(const char* name)
{
    cur_stage = -1;
    for (size_t i = 0; i < stages.size(); ++i)
        if (stages[i].name == name) cur_stage = (int)i;
    if (cur_stage < 0) {
        stages.push_back(ps::StageTimer());
        stages.back().name = name;
        cur_stage = (int)stages.size() - 1;
    }
    cur_start = get_event();
    if (cur_start) (void)hipEventRecord(cur_start, stream);
}

void ps_context::stage_end(int launches)
{
    hipEvent_t stop = get_event();
    if (stop) (void)hipEventRecord(stop, stream);
    if (cur_stage >= 0 && cur_start && stop) {
        stages[cur_stage].spans.push_back({cur_start, stop});
        stages[cur_stage].launches += launches;
    }
    cur_stage = -1;
}

extern "C" {

const char* ps_last_error(void) { return ps::g_err; }

const char* ps_version(void) { return "pointseg-hip 0.1 gfx950"; }

int ps_create(int device, ps_context** out)
{
    PS_CHECK(out != nullptr, "ps_create: out is NULL");
    int count = 0;
    PS_HIP(hipGetDeviceCount(&count));
    PS_CHECK(device >= 0 && device < count, "ps_create: device %d out of range (%d devices)", device, count);
    PS_HIP(hipSetDevice(device));
    ps_context* c = new ps_context();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        ps::set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        return PS_EHIP;
    }
    c->stream = c->own_stream;
    *out = c;
    return PS_OK;
}

int ps_destroy(ps_context* c)
{
    if (!c) return PS_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->knn_arena.buf.release();
    c->net_arena.buf.release();
    c->stage_in.release();
    c->stage_out.release();
    c->ops_ws.release();
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return PS_OK;
}

int ps_set_stream(ps_context* c, void* hip_stream)
{
    PS_CHECK(c != nullptr, "ps_set_stream: ctx is NULL");
    c->stream = static_cast<hipStream_t>(hip_stream);  // NULL = the default (null) stream
    return PS_OK;
}

int ps_synchronize(ps_context* c)
{
    PS_CHECK(c != nullptr, "ps_synchronize: ctx is NULL");
    PS_HIP(hipStreamSynchronize(c->stream));
    return PS_OK;
}

int ps_timing_begin(ps_context* c)
{
    PS_CHECK(c != nullptr, "ps_timing_begin: ctx is NULL");
    c->stages.clear();
    c->event_next = 0;
    c->timing = true;
    return PS_OK;
}

int ps_timing_end(ps_context* c, ps_timing_row* rows, int cap, int* n_rows)
{
    PS_CHECK(c != nullptr && n_rows != nullptr, "ps_timing_end: NULL argument");
    c->timing = false;
    PS_HIP(hipStreamSynchronize(c->stream));
    int n = 0;
    for (auto& st : c->stages) {
        double ms = 0.0;
        for (auto& sp : st.spans) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, sp.first, sp.second) == hipSuccess) ms += t;
        }
        if (rows && n < cap) {
            std::snprintf(rows[n].name, sizeof rows[n].name, "%s", st.name.c_str());
            rows[n].ms = ms;
            rows[n].launches = st.launches;
        }
        ++n;
    }
    *n_rows = n < cap ? n : cap;
    c->stages.clear();
    c->event_next = 0;
    return PS_OK;
}

}  // extern "C"
