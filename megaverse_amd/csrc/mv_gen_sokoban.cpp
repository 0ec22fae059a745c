// megaverse_amd/csrc/mv_gen_sokoban.cpp -- host-side episode generator of the Sokoban scenario.
//
// Replaces SokobanScenario's constructor (level file discovery), reloadLevels, reset, createLayout and the box list of
// addEpisodeDrawables (reference: src/libs/scenarios/src/scenario_sokoban.cpp:38-170,275-293), VoxelGridComponent::toBoundingBoxes
// (component_voxel_grid.hpp:108-187) and the spawn rotation draw of DefaultScenario::spawnAgents (scenario_default.hpp:87).
// Levels are text files in the Boxoban format ('#' wall, '.' goal, '@' player, '+' player on goal, '$' box, '*' box on goal,
// levels separated by lines starting with ';').
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mv_gen.h"

namespace mv {

namespace {

using Rng = std::mt19937;
inline int rand_range(int lo, int hi, Rng &rng) { return std::uniform_int_distribution<>{lo, hi - 1}(rng); }   // util.hpp:30-33
inline float frand01(Rng &rng) { return std::uniform_real_distribution<float>{0, 1}(rng); }                    // util.hpp:46-49

bool slurp(const std::string &path, std::string &out)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    out.clear();
    char buf[4096];
    for (size_t n; (n = std::fread(buf, 1, sizeof buf, f)) > 0;) out.append(buf, n);
    std::fclose(f);
    return true;
}

// util/src/string_utils.cpp:10-25 is strtok_r: runs of separators collapse, empty pieces never appear
std::vector<std::string> nonempty_lines(const std::string &text)
{
    std::vector<std::string> lines;
    size_t at = 0;
    while (at < text.size()) {
        const size_t end = std::min(text.find('\n', at), text.size());
        if (end > at) lines.emplace_back(text, at, end - at);
        at = end + 1;
    }
    return lines;
}

}  // namespace

std::vector<std::string> find_boxoban_level_files()
{
    const char *env = std::getenv("BOXOBAN_LEVELS");
    std::string root = env && *env ? env : "~/datasets/boxoban";
    if (const size_t tilde = root.find('~'); tilde != std::string::npos) {
        const char *home = std::getenv("HOME");
        root.replace(tilde, 1, home ? home : "");
    }
    std::vector<std::string> found;
    for (int i = 0; i <= 999; ++i) {
        char name[16];
        std::snprintf(name, sizeof name, "%03d.txt", i);
        const std::string path = root + "/unfiltered/train/" + name;   // levelSet / levelSplit, scenario_sokoban.hpp:60
        if (FILE *f = std::fopen(path.c_str(), "rb")) { std::fclose(f); found.push_back(path); }
    }
    return found;
}

bool generate_sokoban_episode(std::mt19937 &rng, SokobanLevels &levels, const std::vector<std::string> &files, int num_agents,
                              float base_episode_len, SokobanBlob &out)
{
    std::memset(&out, 0, sizeof out);

    // Env::reset: re-seed from the env's own stream (env.cpp:61-62)
    const int episode_seed = rand_range(0, 1 << 30, rng);
    rng.seed((unsigned long)episode_seed);

    if (levels.pending.empty()) {   // reloadLevels: one random file; a level is kept when the NEXT ';' line shows up
        if (files.empty()) return false;
        std::string text;
        if (!slurp(files[size_t(rand_range(0, int(files.size()), rng))], text)) return false;
        const std::vector<std::string> lines = nonempty_lines(text);
        std::vector<std::string> rows;
        for (size_t i = 0; i < lines.size(); ++i) {
            if (lines[i][0] == ';') {
                if (i > 0) levels.pending.push_back(rows);
                rows.clear();
            } else rows.push_back(lines[i]);
        }
        std::shuffle(levels.pending.begin(), levels.pending.end(), rng);
        if (levels.pending.empty()) return false;
    }
    const std::vector<std::string> rows = levels.pending.back();
    levels.pending.pop_back();

    static const unsigned kFloorColors[5] = {0xffffff, 0xffffe6, 0xe6ecff, 0xffebcc, 0x555555};   // :125-131, env/const.hpp
    out.floor_color = int(kFloorColors[rand_range(0, 5, rng)]);

    // ---- cells: floor under every character of every row, two solid (undrawn) voxels on every '#'
    const int nx = std::min(int(rows.size()), int(SOKO_DIM));
    int nz = 0;
    for (int x = 0; x < nx; ++x) nz = std::max(nz, std::min(int(rows[size_t(x)].size()), int(SOKO_DIM)));
    auto row_len = [&](int x) { return std::min(int(rows[size_t(x)].size()), int(SOKO_DIM)); };
    int agents_placed = 0;
    for (int x = 0; x < nx; ++x)
        for (int z = 0; z < row_len(x); ++z) {
            const char c = rows[size_t(x)][size_t(z)];
            uint8_t &cell = out.cells[x * SOKO_DIM + z];
            if (c == '#') cell = SOKO_WALL;
            if (c == '.' || c == '+') cell = SOKO_GOAL;
            if ((c == '@' || c == '+'))
                for (int k = 0; k < num_agents && agents_placed < MAX_AGENTS; ++k, ++agents_placed) {
                    const float ax = float(x) + float(k % 2) * 0.5f, az = float(z) + float(k % 4 > 1) * 0.5f;
                    out.spawn[agents_placed][0] = ax * 2.0f;
                    out.spawn[agents_placed][1] = float(2.0f + 0.3 * float(k) * 2.0f);   // the reference mixes a double literal in (:155)
                    out.spawn[agents_placed][2] = az * 2.0f;
                }
            if ((c == '$' || c == '*') && out.num_objects < MAX_OBJECTS)
                out.objects[out.num_objects++] = MovableObject{int8_t(x), 1, int8_t(z), 0};
        }
    for (int k = agents_placed; k < num_agents; ++k) std::memcpy(out.spawn[k], out.spawn[0], sizeof out.spawn[0]);   // spawns[0] fallback
    out.dim[0] = nx; out.dim[1] = 3; out.dim[2] = nz;

    // ---- merged slabs: class "solid only" (walls, type 1) before "solid + opaque" (floor, type 3); seeds in (y, z, x) order,
    // grown along x, then z, then y
    {
        const int ny = 3, wz = std::max(nz, 1);
        std::vector<uint8_t> type(size_t(nx) * ny * wz, 0), used(type.size(), 0);
        auto id = [&](int x, int y, int z) { return (size_t(y) * wz + z) * nx + x; };
        for (int x = 0; x < nx; ++x)
            for (int z = 0; z < row_len(x); ++z) {
                type[id(x, 0, z)] = VX_SOLID | VX_OPAQUE;
                if (out.cells[x * SOKO_DIM + z] == SOKO_WALL && rows[size_t(x)][size_t(z)] == '#') type[id(x, 1, z)] = type[id(x, 2, z)] = VX_SOLID;
            }
        for (const int want : {int(VX_SOLID), int(VX_SOLID | VX_OPAQUE)}) {
            auto open_cell = [&](int x, int y, int z) { return x >= 0 && x < nx && y >= 0 && y < ny && z >= 0
                                 && z < wz && type[id(x, y, z)] == want && !used[id(x, y, z)]; };
            for (int y = 0; y < ny; ++y)
                for (int z = 0; z < wz; ++z)
                    for (int x = 0; x < nx; ++x) {
                        if (!open_cell(x, y, z)) continue;
                        int x1 = x + 1, z1 = z + 1, y1 = y + 1;
                        while (open_cell(x1, y, z)) ++x1;
                        auto row_ok = [&](int yy, int zz) { for (int xx = x; xx < x1; ++xx) if (!open_cell(xx, yy, zz)) return false; return true; };
                        while (row_ok(y, z1)) ++z1;
                        auto layer_ok = [&](int yy) { for (int zz = z; zz < z1; ++zz) if (!row_ok(yy, zz)) return false; return true; };
                        while (layer_ok(y1)) ++y1;
                        for (int yy = y; yy < y1; ++yy) for (int zz = z; zz < z1; ++zz) for (int xx = x; xx < x1; ++xx) used[id(xx, yy, zz)] = 1;
                        if (out.num_boxes < MAX_BOXES) {
                            LayoutBox &b = out.boxes[out.num_boxes++];
                            b.min[0] = x; b.min[1] = y; b.min[2] = z; b.max[0] = x1; b.max[1] = y1; b.max[2] = z1;
                            b.type = want; b.slot = 0;
                        }
                    }
        }
    }
    out.episode_len = base_episode_len;
    for (int i = 0; i < num_agents; ++i) out.yaw_frand[i] = frand01(rng);
    return true;
}

}  // namespace mv
