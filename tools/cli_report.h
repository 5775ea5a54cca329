// cli_report.h -- the lines hagrid_cli prints, one function per report of the reference's front-end, writing to any stream so
// that tests/test_cli_report.py can compare them with the reference's own format strings without a GPU:
//   scene    main.cpp:469            "N triangle(s)"
//   grid     main.cpp:512-515        "Grid built in T ms (XxYxZ, C cells, R references)"
//   memory   main.cpp:523-533        "Total memory: / Cells: / Entries: / References: / Triangles: / Peak usage: ... MB"
//   timings  main.cpp:434-444        "I intersection(s)." ... "# Min: T ms"
// Numbers go through operator<< with the stream's default formatting, as in the reference.
#ifndef HAGRID_CLI_REPORT_H
#define HAGRID_CLI_REPORT_H

#include <algorithm>
#include <cstddef>
#include <numeric>
#include <ostream>
#include <vector>

namespace hagrid_cli {

inline void report_scene(std::ostream& os, size_t num_tris) { os << num_tris << " triangle(s)" << std::endl; }

/// built_ms < 0: the grid was read from a file (--load-grid, an extension): "Grid loaded (" instead of the build time.
inline void report_grid(std::ostream& os, double built_ms, int dx, int dy, int dz, int num_cells, int num_refs) {
    if (built_ms >= 0) os << "Grid built in " << built_ms << " ms (";
    else os << "Grid loaded (";
    os << dx << "x" << dy << "x" << dz << ", " << num_cells << " cells, " << num_refs << " references)" << std::endl;
}

inline void report_memory(std::ostream& os, size_t cells_mem, size_t entries_mem, size_t refs_mem, size_t tris_mem, size_t peak) {
    const size_t total_mem = cells_mem + entries_mem + refs_mem + tris_mem;
    os << "Total memory: " << total_mem / double(1024 * 1024) << " MB" << std::endl;
    os << "Cells: " << cells_mem / double(1024 * 1024) << " MB" << std::endl;
    os << "Entries: " << entries_mem / double(1024 * 1024) << " MB" << std::endl;
    os << "References: " << refs_mem / double(1024 * 1024) << " MB" << std::endl;
    os << "Triangles: " << tris_mem / double(1024 * 1024) << " MB" << std::endl;
    os << "Peak usage: " << peak / double(1024.0 * 1024.0) << " MB" << std::endl;
}

/// timings in ms, one per iteration.  The sum is accumulated in single precision, as the reference's
/// std::accumulate(..., 0.0f) does (main.cpp:435), so the same timings print the same digits.
inline void report_timings(std::ostream& os, std::vector<double> timings, size_t rays_per_iter, int intr) {
    const size_t iter = timings.size();
    std::sort(timings.begin(), timings.end());
    const double sum = std::accumulate(timings.begin(), timings.end(), 0.0f);
    const double avg = sum / timings.size();
    const double med = timings[timings.size() / 2];
    const double min = *std::min_element(timings.begin(), timings.end());
    os << intr << " intersection(s)." << std::endl;
    os << sum << "ms for " << iter << " iteration(s)." << std::endl;
    os << rays_per_iter * iter / (1000.0 * sum) << " Mrays/sec." << std::endl;
    os << "# Average: " << avg << " ms" << std::endl;
    os << "# Median: " << med << " ms" << std::endl;
    os << "# Min: " << min << " ms" << std::endl;
}

} // namespace hagrid_cli

#endif // HAGRID_CLI_REPORT_H
