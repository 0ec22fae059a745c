// megaverse_amd/csrc/mv_reset.hip -- episode (re)generation kernel, TowerBuilding.
//
// Replaces, per env:  Env::reset (reference: src/libs/env/src/env.cpp:57-76)
//   -> TowerBuildingScenario::reset + TowerBuildingPlatform::init/generate
//      (src/libs/scenarios/src/scenario_tower_building.cpp:19-89,129-154)
//   -> VoxelGridComponent::addPlatform / toBoundingBoxes (component_voxel_grid.hpp:73-187)
//   -> ObjectStackingComponent::addDrawablesAndCollisions (component_object_stacking.hpp:170-198)
//   -> DefaultScenario::spawnAgents (scenario_default.hpp:80-97)
// and the serial auto-reset loop of VectorEnv::step (src/libs/env/src/vector_env.cpp:93-105),
// which on the GPU is just "the waves whose env is done do work, the others exit".
//
// One wavefront (= one 64-thread block) per env.  The RNG-dependent part is inherently serial and
// runs as uniform scalar code (mv_rng.h); voxel fill and write-out are lane-parallel through a
// 16 KiB LDS image of the chunk so HBM only sees full-width coalesced dwordx4 stores.
#include <hip/hip_runtime.h>

#include "mv_reset_device.h"

namespace mv {

__global__ __launch_bounds__(64) void reset_kernel(GymView gv, int force_all)
{
    const int env = blockIdx.x;
    if (env >= gv.num_envs) return;
    if (!force_all && !gv.hdr[env].done) return;
    reset_env(gv, env, force_all);
}

void launch_reset(const GymView &gv, int force_all, hipStream_t stream)
{
    hipLaunchKernelGGL(reset_kernel, dim3(gv.num_envs), dim3(64), 0, stream, gv, force_all);
}

}  // namespace mv
