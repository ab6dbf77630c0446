// plan_internal.hpp — lets the other translation units append ops to a y6_plan (plan.hip owns the struct).
// A "generic" op is a launcher + a POD descriptor copied into the plan; plan.hip replays it like a conv op.
#pragma once
#include "common.hpp"

typedef int (*y6_generic_fn)(const void* desc, hipStream_t s);

constexpr size_t Y6_GENERIC_BLOB = 1024;

// tag: Y6_TOP_* (include/yolov6_hip.h) for the per-op timing table; flops / bytes: algorithmic work of one launch
int y6_plan_push_generic(y6_plan* p, y6_generic_fn fn, const void* desc, size_t size, int tag, double flops, double bytes);

// the op pushed last writes a caller-visible boundary tensor through the `void*` at byte `offset` of its descriptor
// (y6_plan_rebind_output re-points it)
int y6_plan_mark_output(y6_plan* p, size_t offset);
// ... READS a caller boundary tensor (the NCHW image) through the `const void*` at `offset` (y6_plan_rebind_input / y6_plan_rebind)
int y6_plan_mark_input(y6_plan* p, size_t offset);

template <typename D>
int y6_plan_push(y6_plan* p, int (*fn)(const D*, hipStream_t), const D* d, int tag, double flops, double bytes) {
    static_assert(sizeof(D) <= Y6_GENERIC_BLOB, "descriptor too large for a generic plan op");
    return y6_plan_push_generic(p, reinterpret_cast<y6_generic_fn>(fn), d, sizeof(D), tag, flops, bytes);
}
