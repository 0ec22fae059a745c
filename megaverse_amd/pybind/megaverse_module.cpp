// megaverse_amd/pybind/megaverse_module.cpp -- pybind11 module `megaverse` with the reference's Python-visible table
// (class MegaverseGym + set_megaverse_log_level; reference: src/libs/bindings/megaverse.cpp:267-292), implemented on the
// C ABI of libmegaverse_hip.so (include/megaverse_hip.h).  This is the shim SURVEY.md 8b asks for: drop the built module
// in place of megaverse/extension/megaverse*.so and an unmodified megaverse_env.py keeps working.
//
// Differences a caller can see: errors raise RuntimeError (the reference logs and exit(-1)s); get_observation /
// get_hires_observation return arrays that own a host copy of the frame (the reference returns views into renderer
// memory that the next step() invalidates); draw_overview is the no-GUI build's no-op.  Extra, not in the reference:
// obs_device_ptr() / set_actions_batched() / get_dones() for callers that want the batch without O(num_agents) calls.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "megaverse_hip.h"

namespace py = pybind11;

namespace {

int g_log_level = 2;

void check(int rc)
{
    if (rc < 0) throw std::runtime_error(mv_last_error());
}

// mv_step / mv_reset: 1 = done, with a warning (a capacity limit of this build was hit and reported once; include/megaverse_hip.h)
void check_or_warn(int rc)
{
    check(rc);
    if (rc > 0 && PyErr_WarnEx(PyExc_RuntimeWarning, mv_last_error(), 1) < 0) throw py::error_already_set();
}

int visible_device_count();

class Gym {
public:
    Gym(const std::string &scenario, int w, int h, int num_envs, int num_agents_per_env, int num_simulation_threads, bool use_vulkan,
        const std::map<std::string, float> &float_params)
        : w_{w}, h_{h}, envs_{num_envs}, agents_{num_agents_per_env}
    {
        std::vector<const char *> keys;
        std::vector<float> vals;
        for (const auto &kv : float_params) { keys.push_back(kv.first.c_str()); vals.push_back(kv.second); }
        mv_config cfg{};
        cfg.scenario = scenario.c_str();
        cfg.obs_width = w; cfg.obs_height = h;
        cfg.num_envs = num_envs; cfg.num_agents_per_env = num_agents_per_env;
        cfg.num_simulation_threads = num_simulation_threads; cfg.use_vulkan = use_vulkan ? 1 : 0;
        // The reference's constructor has no device or shard argument (one process = one GPU's worth of envs there too, chosen by
        // CUDA_VISIBLE_DEVICES).  Through this module the process environment decides: MV_DEVICE (taken as is), else LOCAL_RANK (torchrun)
        // modulo the number of devices this process can see -- a launcher that also masks devices per rank (HIP_VISIBLE_DEVICES = one GPU
        // with LOCAL_RANK = k) leaves exactly one, ordinal 0 --, else 0.
        // MV_ENV_OFFSET / MV_TOTAL_ENVS / MV_ENV_STRIDE place this process's envs in a job-wide seed stream (include/megaverse_hip.h:
        // mv_config): a contiguous block, or -- stride k > 1 -- every k-th env from the offset (a multi-task job that deals its scenarios
        // round-robin by env index, one gym per scenario).
        auto env_int = [](const char *name, int fallback) { const char *v = std::getenv(name); return v && *v ? std::atoi(v) : fallback; };
        const int ndev = visible_device_count();
        const int local_rank = env_int("LOCAL_RANK", 0);
        cfg.device = env_int("MV_DEVICE", ndev > 0 && local_rank >= 0 ? local_rank % ndev : 0);
        cfg.env_offset = env_int("MV_ENV_OFFSET", 0);
        cfg.total_envs = env_int("MV_TOTAL_ENVS", 0);
        cfg.env_stride = env_int("MV_ENV_STRIDE", 1);
        cfg.param_keys = keys.data(); cfg.param_vals = vals.data(); cfg.num_params = int(keys.size());
        check(mv_create(&cfg, &gym_));
    }
    ~Gym() { if (gym_) mv_destroy(gym_); }
    Gym(const Gym &) = delete;
    Gym &operator=(const Gym &) = delete;

    int num_agents() const { return mv_num_agents(gym_); }
    std::vector<int> action_space_sizes() const
    {
        int32_t sizes[6];
        check(mv_action_space_sizes(sizes));
        return std::vector<int>(sizes, sizes + 6);
    }
    void seed(int value) { check(mv_seed(gym_, value)); }
    void reset() { check_or_warn(mv_reset(gym_)); }
    void set_actions(int env, int agent, const std::vector<int> &actions)
    {
        std::vector<int32_t> a(actions.begin(), actions.end());
        check(mv_set_actions(gym_, env, agent, a.data(), int(a.size())));
    }
    void step() { check_or_warn(mv_step(gym_)); }
    bool is_done(int env)
    {
        const int rc = mv_is_done(gym_, env);
        check(rc);
        return rc != 0;
    }
    py::array_t<uint8_t> get_observation(int env, int agent)
    {
        py::array_t<uint8_t> frame({h_, w_, 4});
        check(mv_get_observation(gym_, env, agent, frame.mutable_data()));
        return frame;
    }
    std::vector<float> get_last_rewards()
    {
        std::vector<float> rewards(size_t(envs_) * agents_);
        check(mv_get_last_rewards(gym_, rewards.data()));
        return rewards;
    }
    float true_objective(int env, int agent)
    {
        float v = 0.0f;
        check(mv_true_objective(gym_, env, agent, &v));
        return v;
    }
    void set_render_resolution(int w, int h)
    {
        check(mv_set_render_resolution(gym_, w, h));
        render_w_ = w; render_h_ = h;
    }
    void draw_hires() { check(mv_draw_hires(gym_)); }
    void draw_overview() { check(mv_draw_overview(gym_)); }
    py::array_t<uint8_t> get_hires_observation(int env, int agent)
    {
        py::array_t<uint8_t> frame({render_h_, render_w_, 4});
        check(mv_get_hires_observation(gym_, env, agent, frame.mutable_data()));
        return frame;
    }
    std::map<std::string, float> get_reward_shaping(int env, int agent)
    {
        std::map<std::string, float> out;
        for (int i = 0; i < mv_num_reward_shaping_keys(gym_); ++i) {
            const char *key = mv_reward_shaping_key(gym_, i);
            float v = 0.0f;
            check(mv_get_reward_shaping(gym_, env, agent, key, &v));
            out[key] = v;
        }
        return out;
    }
    void set_reward_shaping(int env, int agent, const std::map<std::string, float> &shaping)
    {
        for (const auto &kv : shaping) check(mv_set_reward_shaping(gym_, env, agent, kv.first.c_str(), kv.second));
    }
    void close() { if (gym_) check(mv_close(gym_)); }

    // ---- additions
    void set_actions_batched(py::array_t<int32_t, py::array::c_style | py::array::forcecast> actions)
    {
        if (actions.size() != py::ssize_t(envs_) * agents_ * 6) throw std::runtime_error("set_actions_batched: expected num_envs * num_agents_per_env * 6 values");
        check(mv_set_actions_batched(gym_, actions.data()));
    }
    py::array_t<uint8_t> get_dones()
    {
        py::array_t<uint8_t> dones(envs_);
        check(mv_get_dones(gym_, dones.mutable_data()));
        return dones;
    }
    std::uintptr_t obs_device_ptr() const { return reinterpret_cast<std::uintptr_t>(mv_obs_device_ptr(gym_)); }

private:
    mv_gym *gym_ = nullptr;
    int w_, h_, envs_, agents_, render_w_ = 768, render_h_ = 432;   // default hires size: megaverse.cpp:261
};

int visible_device_count() { return mv_device_count(); }

}  // namespace

PYBIND11_MODULE(megaverse, m)
{
    m.doc() = "Megaverse Python bindings (MI355X HIP back end)";
    m.def("set_megaverse_log_level", [](int level) { g_log_level = level; }, "accepted for API parity: the HIP library does not log");
    py::class_<Gym>(m, "MegaverseGym")
        .def(py::init<const std::string &, int, int, int, int, int, bool, const std::map<std::string, float> &>())
        .def("num_agents", &Gym::num_agents)
        .def("action_space_sizes", &Gym::action_space_sizes)
        .def("seed", &Gym::seed)
        .def("reset", &Gym::reset)
        .def("set_actions", &Gym::set_actions)
        .def("step", &Gym::step)
        .def("is_done", &Gym::is_done)
        .def("get_observation", &Gym::get_observation)
        .def("get_last_rewards", &Gym::get_last_rewards)
        .def("true_objective", &Gym::true_objective)
        .def("set_render_resolution", &Gym::set_render_resolution)
        .def("draw_hires", &Gym::draw_hires)
        .def("draw_overview", &Gym::draw_overview)
        .def("get_hires_observation", &Gym::get_hires_observation)
        .def("get_reward_shaping", &Gym::get_reward_shaping)
        .def("set_reward_shaping", &Gym::set_reward_shaping)
        .def("close", &Gym::close)
        .def("set_actions_batched", &Gym::set_actions_batched)
        .def("get_dones", &Gym::get_dones)
        .def("obs_device_ptr", &Gym::obs_device_ptr);
}
