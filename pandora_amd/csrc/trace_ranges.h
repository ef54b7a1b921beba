// Named profiler regions (roctx) around the host side of the path: upload, every round's top-level episode / binning / bottom
// level, leaf streaming and the exchanges of the sharded loop. The reference wraps the same points with Optick events
// (stream_framework/include/stream/task_graph.h:249-259: static-data load and kernel execute;
// offline_batching_acceleration_structure.h:165-175). `rocprofv3 --marker-trace --kernel-trace` shows them next to the kernels.
// The roctx library is looked up at run time (it is part of ROCm, but a host that only links libpandora_hip.so must not need it):
// without it -- or with PANDORA_ROCTX=0 -- a range costs one predictable branch.
#pragma once

namespace pandora_hip {

struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
const RoctxApi& roctxApi(); // abi.hip

struct TraceRange {
    bool on;
    explicit TraceRange(const char* name)
        : on(roctxApi().push != nullptr)
    {
        if (on)
            roctxApi().push(name);
    }
    ~TraceRange()
    {
        if (on)
            roctxApi().pop();
    }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};

} // namespace pandora_hip
