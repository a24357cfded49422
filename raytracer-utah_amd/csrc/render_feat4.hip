// render_feat4.hip — the kernels of render_impl.h instantiated for feature set 4 (frames in flight, untextured).
// One translation unit per feature set so that the library builds in parallel (make -j).
#include "render_impl.h"

int rtu_launch_feat4(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe) {
    if (bvh_stack_needed <= 16) return launch_all<16, 4>(args, n_tiles, stats, stream, mode, probe);
    if (bvh_stack_needed <= 24) return launch_all<24, 4>(args, n_tiles, stats, stream, mode, probe);
    if (bvh_stack_needed <= 32) return launch_all<32, 4>(args, n_tiles, stats, stream, mode, probe);
    return launch_all<RTU_MAX_BVH_STACK, 4>(args, n_tiles, stats, stream, mode, probe);
}
