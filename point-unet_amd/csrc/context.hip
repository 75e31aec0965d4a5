// context.hip -- ps_context lifetime, error text, workspace buffers, hipEvent stage timing.
#include "common.h"

#include <cstdlib>

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

int ps_context::check_flag_slot(int s)
{
    int rc = PS_OK;
    if (pending_mask & (1u << s)) {
        const int32_t* f = h_flags + 4 * s;
        // (f[2], the "unfinished build" word of the first builder, is always 0: the straggler kernel finishes every tree)
        if (f[1] != 0) {
            ps::set_error("deferred check: pyramid build #%llu of this context: kd-tree builder queue overflow (degenerate cloud); its index "
                          "tables were filled with index 0", (unsigned long long)flag_serial[s]);
            rc = PS_ESTATE;
        } else if (f[0] != 0) {
            ps::set_error("deferred check: pyramid build #%llu of this context: kd-tree deeper than the traversal stack (degenerate cloud); "
                          "its neighbour lists are valid indices but may not be the nearest", (unsigned long long)flag_serial[s]);
            rc = PS_ESTATE;
        }
        pending_mask &= ~(1u << s);
    }
    return rc;
}

int ps_context::check_deferred()
{
    int rc = PS_OK;
    for (int s = 0; s < 8; ++s) {
        const int r = check_flag_slot(s);
        if (r != PS_OK) rc = r;
    }
    if (sticky_rc != PS_OK) {  // the oldest failure wins: it was found first (ps_pyramid_build, slot reuse)
        ps::set_error("%s", sticky_msg.c_str());
        rc = sticky_rc;
        sticky_rc = PS_OK;
        sticky_msg.clear();
    }
    return rc;
}

int ps_context::upload_async(void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return PS_OK;
    PinSlot& s = pin[pin_pos];
    pin_pos = (pin_pos + 1) & 15;
    if (s.busy) {
        PS_HIP(hipEventSynchronize(s.ev));  // the copy that last used this slot has drained
        s.busy = false;
    }
    if (s.cap < bytes) {
        if (s.p) (void)hipHostFree(s.p);
        s.p = nullptr;
        s.cap = 0;
        const size_t want = bytes < 16384 ? 16384 : bytes + bytes / 4;
        PS_HIP(hipHostMalloc(&s.p, want));
        s.cap = want;
    }
    if (!s.ev) PS_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    std::memcpy(s.p, src, bytes);
    PS_HIP(hipMemcpyAsync(dst, s.p, bytes, hipMemcpyHostToDevice, stream));
    PS_HIP(hipEventRecord(s.ev, stream));
    s.busy = true;
    return PS_OK;
}

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

}  // extern "C"

namespace ps {
void tuning_from_env(Tuning& t)
{
#ifdef PS_TUNING_ENV
    auto num = [](const char* name, double& out) { const char* v = std::getenv(name); if (v && *v) { out = std::atof(v); return true; } return false; };
    auto flag = [&](const char* name, bool& b) { double v; if (num(name, v)) b = v != 0; };
    auto clampi = [&](const char* name, int64_t lo, int64_t hi, int64_t& x) { double v; if (num(name, v)) x = v < lo ? lo : (v > hi ? hi : (int64_t)v); };
    double v;
    if (num("PS_GEMM32B_MIN_FLOPS", v) && v >= 0) t.gemm32b_min_flops = v;
    if (num("PS_GEMM32B_RW", v) && (v == 1 || v == 2)) t.gemm32b_rw = (int)v;
    if (num("PS_GEMM32B_CW", v) && (v == 1 || v == 2)) t.gemm32b_cw = (int)v;
    t.gemm32_no_sk8 = std::getenv("PS_GEMM32_NO_SK8") != nullptr;
    flag("PS_ATT64_GEMM", t.att64_gemm);
    if (num("PS_ATT64_OCC", v) && (v == 1 || v == 2)) t.att64_occ = (int)v;
    t.att_no_split = std::getenv("PS_ATT_NO_SPLIT") != nullptr;
    clampi("PS_WGRAD_B3_MIN_ROWS", 0, 1ll << 40, t.wgrad_b3_min_rows);
    clampi("PS_GEMM_B3_MIN_ROWS", 0, 1ll << 40, t.gemm_b3_min_rows);
    clampi("PS_WGRAD_WGS", 64, 4096, t.wgrad_wgs);
    flag("PS_BN_SLICE", t.bn_slice);
    flag("PS_INV_BUCKET", t.inv_bucket);
    if (num("PS_INV_TILE", v) && (v == 4096 || v == 8192)) t.inv_tile = (int)v;
    flag("PS_GATHER_REDUCE_ORDERED", t.gather_reduce_ordered);
    if (num("PS_MAXPOOL_BWD_ORDERED", v)) t.maxpool_bwd_ordered = v != 0;
    flag("PS_TRAIN_ACT_BF16", t.train_act_bf16);
    if (num("PS_TRAIN_ATT_GEMM_SPLIT", v)) t.train_att_gemm_split = v < 0 ? -1 : (v != 0);
    flag("PS_TRAIN_ATT_GEMM", t.train_att_gemm);
    flag("PS_TRAIN_ATT128_FWD_GEMM", t.train_att128_fwd_gemm);
    flag("PS_TRAIN_FUSE_RESIDUAL", t.train_fuse_residual);
    flag("PS_TRAIN_MERGE_SYNCBN", t.train_merge_syncbn);
    if (num("PS_CONVBN_MAX_C", v) && v >= 0 && v <= 4096) t.convbn_max_c = (int)v;
    if (num("PS_CONVBN_RECT_MAX", v) && v >= 0) t.convbn_rect_max = v > (double)(1 << 30) ? 1 << 30 : (int)v;
    t.wgrad_debug = std::getenv("PS_WGRAD_DEBUG") != nullptr;
#else
    (void)t;
#endif
}
}  // namespace ps

extern "C" {
const char* ps_version(void) { return "pointseg-hip 0.1 gfx950"; }
int ps_abi_version(void) { return PS_ABI_VERSION; }

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
    ps::tuning_from_env(c->tune);  // (nothing in the default build: common.h, struct Tuning)
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
    c->red_ws.release();
    c->wgrad_ws.release();
    for (auto& b : c->ops_ring) b.release();
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    for (auto& e : c->flag_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& s : c->pin) {
        if (s.p) (void)hipHostFree(s.p);
        if (s.ev) (void)hipEventDestroy(s.ev);
    }
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
    return c->check_deferred();
}

int ps_set_train_gemm_bf16(ps_context* c, int on)
{
    PS_CHECK(c != nullptr, "ps_set_train_gemm_bf16: ctx is NULL");
    c->train_bf16 = on != 0;
    return PS_OK;
}

int ps_set_train_act_bf16(ps_context* c, int on)
{
    PS_CHECK(c != nullptr, "ps_set_train_act_bf16: ctx is NULL");
    c->train_act_bf16 = on != 0;
    return PS_OK;
}

int ps_set_train_gemm_b3(ps_context* c, int on)
{
    PS_CHECK(c != nullptr, "ps_set_train_gemm_b3: ctx is NULL");
    c->train_b3 = on != 0;
    return PS_OK;
}

int ps_set_att_bf16x3(ps_context* c, int on)
{
    PS_CHECK(c != nullptr, "ps_set_att_bf16x3: ctx is NULL");
    c->att_bf16x3 = on != 0;
    return PS_OK;
}

int ps_set_deferred_checks(ps_context* c, int on)
{
    PS_CHECK(c != nullptr, "ps_set_deferred_checks: ctx is NULL");
    if (on && !c->h_flags) PS_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_flags), 8 * 4 * sizeof(int32_t)));
    if (!on && c->pending_mask) {
        PS_HIP(hipStreamSynchronize(c->stream));
        PS_TRY(c->check_deferred());
    }
    c->deferred = on != 0;
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

int ps_timing_select(ps_context* c, const char* stage_name)
{
    PS_CHECK(c != nullptr, "ps_timing_select: ctx is NULL");
    c->timing_only = stage_name ? stage_name : "";
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
