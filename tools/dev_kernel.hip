// ISA study of ONE instantiation of the step kernel (tuning only; not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize --offload-device-only -S tools/dev_kernel.hip -o /tmp/k.s \
//         -DSCG_DEV_ARGS="SCG_QUAD_3D, 0, scg::FT_TRACK, scg::MODE_SINGLE"
//   python3 tools/isa_stats.py /tmp/k.s scg_step_kernel
#define SCG_NO_LAUNCHERS 1
#include "../safe_control_gym_amd/csrc/scg_kernels.hip"
#ifndef SCG_DEV_ARGS
#define SCG_DEV_ARGS SCG_QUAD_3D, 0, scg::FT_TRACK, scg::MODE_SINGLE
#endif
template __global__ void scg::scg_step_kernel<SCG_DEV_ARGS>(const scg::KParams, const scg_step_io, const scg::KTables*,
                                                            const scg::ConTable*, const float*, const uint32_t,
                                                            const scg::RolloutArgs);
