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

// mv_reset: every env takes the next episode of its ring (tower_draw_kernel has run in front of this launch: mv_api.hip)
__global__ __launch_bounds__(64) void reset_kernel(GymView gv, int force_all)
{
    const int env = blockIdx.x;
    if (env >= gv.num_envs) return;
    if (!force_all && !gv.hdr[env].done) return;
    (void)tower_swap_in(gv, env, force_all);
}

// The episode generator's serial half, one wavefront per env: tops the env's ring of resident episodes up to `spares` beyond what it has consumed.
// Almost every wave finds its ring full and leaves after three loads; the few whose env finished since the last launch work ~47 us each -- on a
// stream of their own, beside the step kernels and the observation passes (mv_api.hip: tower_draw_after).
__global__ __launch_bounds__(64) void tower_draw_kernel(GymView gv)
{
    const int env = blockIdx.x;
    if (env >= gv.num_envs) return;
    const int consumed = __builtin_amdgcn_readfirstlane(gv.hdr[env].episodes_consumed);
    int generated = __builtin_amdgcn_readfirstlane(gv.tower_gen[env].generated);
    while (generated < consumed + gv.spares) {
        ++generated;
        tower_draw(gv, env, generated);
    }
}

// Env::seed (env.cpp:52-55) for every env: the generator restarts from the new seed, what was drawn ahead from the old stream is dropped
__global__ void tower_seed_kernel(GymView gv, const uint32_t *seeds)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= gv.num_envs) return;
    TowerGen tg;
    tg.seed = seeds[env]; tg.seed_is_env_seed = 1; tg.generated = gv.hdr[env].episodes_consumed; tg.pad = 0;
    gv.tower_gen[env] = tg;
    TowerBlob *ring = const_cast<TowerBlob *>(reinterpret_cast<const TowerBlob *>(gv.blobs)) + (size_t)env * gv.spares;
    for (int q = 0; q < gv.spares; ++q) ring[q].seq = 0;
}

void launch_reset(const GymView &gv, int force_all, hipStream_t stream)
{
    hipLaunchKernelGGL(reset_kernel, dim3(gv.num_envs), dim3(64), 0, stream, gv, force_all);
}

void launch_tower_draw(const GymView &gv, hipStream_t stream)
{
    hipLaunchKernelGGL(tower_draw_kernel, dim3(gv.num_envs), dim3(64), 0, stream, gv);
}

void launch_tower_seed(const GymView &gv, const uint32_t *seeds, hipStream_t stream)
{
    hipLaunchKernelGGL(tower_seed_kernel, dim3((gv.num_envs + 255) / 256), dim3(256), 0, stream, gv, seeds);
}

}  // namespace mv
