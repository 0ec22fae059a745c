/*
 * oracle/ref_mazes_shim.cpp -- C wrapper around the REFERENCE's own maze library (src/libs/mazes, vendored and self-contained).
 *
 * TEST INFRASTRUCTURE.  This file contains no reference code: oracle/Makefile compiles the library's sources where they lie under
 * /root/reference/src/libs/mazes (honeycombmaze.cpp, maze.cpp, kruskal.cpp, spanningtreealgorithm.cpp, cellborder.cpp) together with
 * this wrapper into oracle/_ref/libmv_ref_mazes.so.  The library seeds its Kruskal from std::random_device; SeededKruskal re-seeds the
 * (protected) generator so that a maze can be reproduced and compared with the oracle's restatement (mvo_hex_maze).
 */
#include <mazes/honeycombmaze.h>
#include <mazes/kruskal.h>

namespace {
struct SeededKruskal : Kruskal {
    explicit SeededKruskal(unsigned seed) { generator.seed(seed); }
};
}

extern "C" int mvref_hex_maze(int size, unsigned seed, int *cells_out, int *border_counts, int *border_to, double *border_xy, double *centers,
                              double *bounds)
{
    HoneyCombMaze maze(size);
    SeededKruskal algorithm(seed);
    maze.InitialiseGraph();
    maze.GenerateMaze(&algorithm);
    auto &adj = maze.getAdjacencyList();
    if (cells_out) *cells_out = int(adj.size());
    int n = 0;
    for (size_t i = 0; i < adj.size(); ++i) {
        if (border_counts) border_counts[i] = int(adj[i].size());
        for (auto &entry : adj[i]) {
            if (border_to) border_to[n] = entry.first;
            if (border_xy) {
                const auto xy = dynamic_cast<LineBorder *>(entry.second.get())->getBorderCoords();
                border_xy[4 * n] = std::get<0>(xy); border_xy[4 * n + 1] = std::get<1>(xy); border_xy[4 * n + 2] = std::get<2>(xy); border_xy[4 * n + 3] = std::get<3>(xy);
            }
            ++n;
        }
        if (centers) { centers[2 * i] = maze.getCellCenters()[i].first; centers[2 * i + 1] = maze.getCellCenters()[i].second; }
    }
    if (bounds) { const auto b = maze.GetCoordinateBounds(); bounds[0] = std::get<0>(b); bounds[1] = std::get<1>(b); bounds[2] = std::get<2>(b); bounds[3] = std::get<3>(b); }
    return n;
}
