// megaverse_amd/csrc/mv_agents.h -- the agents of one env during a tick: records in LDS, one agent's physics fields in registers at a time.
//
// The step kernels used to keep AgentState ag[A_MAX] in VGPRs with every loop over agents unrolled: 272-1040 bytes of scratch per lane
// at 2-8 agents and a tick of 66 us at 512 envs x 4 agents against 25 us at one agent.  Now the env's wavefront copies the
// A records (128 B each) into LDS, every loop over agents is a real loop, the controller works on one agent's 15 physics values in
// registers (wave-uniform), scenario logic reads and writes the few fields it needs in place, and rewardTeam is one lane per agent.
// Float operation order per agent is unchanged (env.cpp:85-152, scenario.hpp:259-298): results stay bit-identical to the oracle.
#pragma once
#include <hip/hip_runtime.h>

#include "mv_actions.h"
#include "mv_math.h"
#include "mv_types.h"

namespace mv {
namespace {

enum : int { AGENT_DWORDS = sizeof(AgentState) / 4, AGENT_DYNAMIC_DWORDS = 23 };   // dwords a tick can change: pos .. total_reward
static_assert(offsetof(AgentState, shaping) == AGENT_DYNAMIC_DWORDS * 4, "AgentState layout changed: update AGENT_DYNAMIC_DWORDS");

// global -> LDS, this tick's actions, last_reward = 0 (env.cpp:85).  Called by the env's whole wavefront.
__device__ __forceinline__ void agents_load(const GymView &gv, int env, int A, AgentState *s_ag, int *s_act)
{
    const int lane = lane_id();
    const uint32_t *src = reinterpret_cast<const uint32_t *>(gv.agents + (size_t)env * A);
    uint32_t *dst = reinterpret_cast<uint32_t *>(s_ag);
    for (int i = lane; i < A * AGENT_DWORDS; i += 64) dst[i] = src[i];
    if (lane < A) s_act[lane] = action_of(gv, env, lane);
    wave_sync();
    if (lane < A) s_ag[lane].last_reward = 0.0f;
    wave_sync();
}

// LDS -> global: the fields a tick can change, the reported rewards, the cleared actions (env.cpp:141-142)
__device__ __forceinline__ void agents_store(const GymView &gv, int env, int A, AgentState *s_ag)
{
    const int lane = lane_id();
    wave_sync();
    if (lane < A) s_ag[lane].total_reward += s_ag[lane].last_reward;
    wave_sync();
    const uint32_t *src = reinterpret_cast<const uint32_t *>(s_ag);
    uint32_t *dst = reinterpret_cast<uint32_t *>(gv.agents + (size_t)env * A);
    for (int i = lane; i < A * AGENT_DWORDS; i += 64)
        if ((i & (AGENT_DWORDS - 1)) < AGENT_DYNAMIC_DWORDS) dst[i] = src[i];
    if (lane < A) {
        gv.actions[(size_t)env * A + lane] = 0;
        gv.rewards[(size_t)env * A + lane] = s_ag[lane].last_reward;   // zeroed by the reset if the env is done
    }
}

// the 15 values the character controller, the look/accelerate/jump actions and the camera work on
__device__ __forceinline__ void phys_load(AgentState &a, const AgentState &s)
{
    a.pos[0] = s.pos[0]; a.pos[1] = s.pos[1]; a.pos[2] = s.pos[2];
    a.m00 = s.m00; a.m02 = s.m02; a.m20 = s.m20; a.m22 = s.m22; a.pitch = s.pitch;
    a.hvx = s.hvx; a.hvz = s.hvz; a.vvel = s.vvel; a.voffset = s.voffset; a.step_offset = s.step_offset; a.jump_speed = s.jump_speed;
    a.was_jumping = s.was_jumping;
}
__device__ __forceinline__ void phys_store(AgentState &s, const AgentState &a)
{
    s.pos[0] = a.pos[0]; s.pos[1] = a.pos[1]; s.pos[2] = a.pos[2];
    s.m00 = a.m00; s.m02 = a.m02; s.m20 = a.m20; s.m22 = a.m22; s.pitch = a.pitch;
    s.hvx = a.hvx; s.hvz = a.hvz; s.vvel = a.vvel; s.voffset = a.voffset; s.step_offset = a.step_offset; s.jump_speed = a.jump_speed;
    s.was_jumping = a.was_jumping;
}

// Scenario::rewardAgent / rewardTeam (scenario.hpp:259-298) on the LDS records: the actor's lane adds its individual share, then every
// agent's lane adds the team share -- the actor's two additions stay in program order, the others are independent
__device__ __forceinline__ void reward_agent_lds(AgentState *s_ag, int key, int idx, float mult)
{
    if (lane_id() == idx) s_ag[idx].last_reward += s_ag[idx].shaping[key] * mult;
    wave_sync();
}
__device__ __forceinline__ void reward_team_lds(AgentState *s_ag, int A, int key, int idx, float mult)
{
    const int lane = lane_id();
    if (lane == idx) s_ag[lane].last_reward += s_ag[lane].shaping[key] * (mult * (1 - s_ag[lane].shaping[0]));
    if (lane < A) s_ag[lane].last_reward += s_ag[lane].shaping[key] * s_ag[lane].shaping[0] * mult / float(A);
    wave_sync();
}

}  // namespace
}  // namespace mv
