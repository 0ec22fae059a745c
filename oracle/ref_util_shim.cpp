/*
 * oracle/ref_util_shim.cpp -- C wrappers around the REFERENCE's own RNG helpers.
 *
 * TEST INFRASTRUCTURE.  This file contains no reference code: it #includes
 * <util/util.hpp> from where it lies under /root/reference/src/libs/util/include
 * (-I given by oracle/Makefile) and exports randRange / randomBool / frand
 * (util.hpp:25-56) through a C ABI so tests can pin oracle/ and the HIP RNG
 * restatement against the real reference functions.  Output goes to oracle/_ref/
 * (git-ignored, travels to the GPU box with the snapshot).
 *
 * Only util.hpp (+ macro.hpp), math_utils.hpp, string_utils.{hpp,cpp} and the self-contained perlin_noise.hpp are buildable
 * this way: every other header on the hot path pulls in Magnum/Corrade/Bullet, which are not
 * vendored -- env/const.hpp (colours, parameter names) through util/magnum.hpp,
 * env/voxel_state.hpp through env/physics.hpp (btBulletDynamicsCommon.h), scenarios/const.hpp
 * (reward-shaping key strings) through env/const.hpp.  Writing stand-ins for those would be
 * faking the reference build; their constants stay restated by hand and are checked against the
 * cited lines by tests/test_oracle_spec.py.
 */
#include <util/util.hpp>
#include <util/math_utils.hpp>
#include <util/perlin_noise.hpp>
#include <util/string_utils.hpp>   // splitString: compiled from the reference's util/src/string_utils.cpp (oracle/Makefile)

#include <cstring>

extern "C" {

void mvref_rand_range_seq(unsigned seed, const int *lo, const int *hi, int n, int *out)
{
    Megaverse::Rng rng(seed);
    for (int i = 0; i < n; ++i) out[i] = Megaverse::randRange(lo[i], hi[i], rng);
}

void mvref_frand_seq(unsigned seed, int n, float *out)
{
    Megaverse::Rng rng(seed);
    for (int i = 0; i < n; ++i) out[i] = Megaverse::frand(rng);
}

void mvref_random_bool_seq(unsigned seed, int n, int *out)
{
    Megaverse::Rng rng(seed);
    for (int i = 0; i < n; ++i) out[i] = Megaverse::randomBool(rng) ? 1 : 0;
}

/* MegaverseGym::seed's per-env seed rule restated with the reference helper:
 * bindings/megaverse.cpp:64-68 */
void mvref_env_seeds(int seed, int n, int *out)
{
    Megaverse::Rng rng;
    rng.seed((unsigned long)seed);
    for (int i = 0; i < n; ++i) out[i] = Megaverse::randRange(0, 1 << 30, rng);
}

/* triangularNumber (util/math_utils.hpp:7-10): how many boxes an Obstacles wall / gap of a given size needs (platforms.hpp) */
int mvref_triangular_number(int n) { return Megaverse::triangularNumber(n); }

/* CollectScenario::createLandscape's noise call (scenario_collect.cpp:80,88) on the reference's vendored
 * siv::PerlinNoise (util/perlin_noise.hpp:118-126,315-318) */
void mvref_perlin_octave2_01(unsigned seed, const double *xs, const double *ys, int n, int octaves, double *out)
{
    const siv::PerlinNoise perlin(seed);
    for (int i = 0; i < n; ++i) out[i] = perlin.accumulatedOctaveNoise2D_0_1(xs[i], ys[i], octaves);
}


/* splitString (util/src/string_utils.cpp:10-25, strtok_r: runs of separators collapse), as SokobanScenario::reloadLevels uses it on a
 * level file (scenario_sokoban.cpp:90): the tokens joined by '\x1f' into out (at most cap bytes incl. the terminator); returns their number */
int mvref_split_string(const char *text, const char *delims, char *out, int cap)
{
    const std::vector<std::string> tokens = Megaverse::splitString(text, delims);
    std::string joined;
    for (size_t i = 0; i < tokens.size(); ++i) { if (i) joined += '\x1f'; joined += tokens[i]; }
    if (cap > 0) { std::strncpy(out, joined.c_str(), size_t(cap) - 1); out[cap - 1] = 0; }
    return int(tokens.size());
}
}
