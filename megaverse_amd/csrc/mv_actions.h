// megaverse_amd/csrc/mv_actions.h -- action encoding shared by the host-facing kernels and the step kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mv_types.h"

namespace mv {
namespace {

// MegaverseGym::setActions, bindings/megaverse.cpp:100-116: multi-discrete [6] (sizes 3,3,3,2,2,3) -> Action bitmask
__device__ __forceinline__ int action_mask_of(const int32_t *a)
{
    int idx = 0, mask = 0;
    const int sizes[6] = {3, 3, 3, 2, 2, 3};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        if (a[i] > 0) mask |= 1 << (idx + a[i]);
        idx += sizes[i] - 1;
    }
    return mask;
}

__device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// The benchmark's random policy (= action_space.sample(), megaverse_env.py:110-112): i.i.d. uniform per head from a
// counter-based generator keyed on (seed, step, JOB-WIDE agent id), so that a sharded or strided gym draws exactly the
// actions the single big gym would; megaverse_amd/rollout.py:sample_actions is the host twin.
__device__ __forceinline__ int sampled_action_mask(const GymView &gv, int env, int agent)
{
    const uint32_t gid = (uint32_t)(gv.env_offset + env * gv.env_stride) * (uint32_t)gv.num_agents + (uint32_t)agent;
    const uint32_t base = fmix32(fmix32(gv.sample_seed ^ fmix32(gv.sample_step + 0x9E3779B9u)) ^ (gid * 0x85EBCA6Bu + 1u));
    const int sizes[6] = {3, 3, 3, 2, 2, 3};
    int32_t a[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t hsh = fmix32(base + (uint32_t)k * 0xC2B2AE35u);
        a[k] = (int32_t)(((uint64_t)hsh * (uint64_t)sizes[k]) >> 32);
    }
    return action_mask_of(a);
}

// The reference's own benchmark policy (src/apps/megaverse_test_app.cpp:140-147): Action(1 << randRange(0, int(Action::NumActions))) per agent
// and tick, NumActions = 11 (env.hpp:22-42; bit 0 is no action at all).  Same counter-based generator and keying as above; the host twin is
// megaverse_amd/rollout.py:sample_single_bit_masks.
__device__ __forceinline__ int sampled_single_bit_mask(const GymView &gv, int env, int agent)
{
    const uint32_t gid = (uint32_t)(gv.env_offset + env * gv.env_stride) * (uint32_t)gv.num_agents + (uint32_t)agent;
    const uint32_t base = fmix32(fmix32(gv.sample_seed ^ fmix32(gv.sample_step + 0x9E3779B9u)) ^ (gid * 0x85EBCA6Bu + 1u));
    const uint32_t hsh = fmix32(base + 6u * 0xC2B2AE35u);   // (a head index the multi-discrete policy does not use)
    return 1 << (int)(((uint64_t)hsh * 11ull) >> 32);
}

// the action mask agent `agent` of env `env` acts on this tick
__device__ __forceinline__ int action_of(const GymView &gv, int env, int agent)
{
    if (gv.sample_on == POLICY_MULTIDISCRETE) return sampled_action_mask(gv, env, agent);
    if (gv.sample_on == POLICY_SINGLE_BIT) return sampled_single_bit_mask(gv, env, agent);
    const size_t i = (size_t)env * gv.num_agents + agent;
    return gv.md_actions ? action_mask_of(gv.md_actions + i * 6) : gv.actions[i];
}

}  // namespace
}  // namespace mv
