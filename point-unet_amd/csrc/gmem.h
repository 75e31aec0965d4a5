// gmem.h -- loads/stores with an explicit GLOBAL address space.
// Pointers that reach a kernel through a table in memory (job / tree descriptors) are "generic" to the compiler, which
// then emits flat_load / flat_store: those are tracked by both the vector-memory and the LDS counters, so every wait
// becomes `s_waitcnt vmcnt(0) lgkmcnt(0)` and loads cannot overlap LDS traffic.  Everything such a descriptor points
// to is device global memory; these helpers say so and get global_load / global_store.  (Host pass: plain accesses.)
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PS_GM __host__ __device__ __forceinline__
#else
#define PS_GM inline
#endif

namespace ps {

#if defined(__HIP_DEVICE_COMPILE__)
#define PS_AS1 __attribute__((address_space(1)))
typedef int gm_v4i __attribute__((ext_vector_type(4)));
typedef int gm_v2i __attribute__((ext_vector_type(2)));
typedef float gm_v4f __attribute__((ext_vector_type(4)));

PS_GM int4 gload(const int4* p)
{
    const gm_v4i v = *(const PS_AS1 gm_v4i*)p;
    return make_int4(v.x, v.y, v.z, v.w);
}
PS_GM int2 gload(const int2* p)
{
    const gm_v2i v = *(const PS_AS1 gm_v2i*)p;
    return make_int2(v.x, v.y);
}
PS_GM float4 gload(const float4* p)
{
    const gm_v4f v = *(const PS_AS1 gm_v4f*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
PS_GM float gload(const float* p) { return *(const PS_AS1 float*)p; }
PS_GM int gload(const int* p) { return *(const PS_AS1 int*)p; }
PS_GM unsigned gload(const unsigned* p) { return *(const PS_AS1 unsigned*)p; }

PS_GM void gstore(int4* p, int4 v)
{
    gm_v4i t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *(PS_AS1 gm_v4i*)p = t;
}
PS_GM void gstore(float4* p, float4 v)
{
    gm_v4f t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *(PS_AS1 gm_v4f*)p = t;
}
PS_GM void gstore(float* p, float v) { *(PS_AS1 float*)p = v; }
PS_GM void gstore(int* p, int v) { *(PS_AS1 int*)p = v; }
PS_GM uint8_t gload(const uint8_t* p) { return *(const PS_AS1 uint8_t*)p; }
PS_GM void gstore(uint8_t* p, uint8_t v) { *(PS_AS1 uint8_t*)p = v; }
PS_GM void gstore(unsigned* p, unsigned v) { *(PS_AS1 unsigned*)p = v; }

#else

template <class T>
PS_GM T gload(const T* p) { return *p; }
template <class T, class U>
PS_GM void gstore(T* p, U v) { *p = v; }

#endif

}  // namespace ps
