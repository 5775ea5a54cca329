// hagrid_cli -- command-line front-end over include/hagrid/*.h with the option names, defaults and output lines of
// the reference's `hagrid` executable (src/main.cpp:113-244 options, :469 / :512-533 build report, :434-444 benchmark
// report), minus the SDL viewer: without a ray file it traces ONE frame of primary rays -- the viewer's initial view, gen_camera / gen_rays formulas
// (main.cpp:42-66, :572-598) -- and can write it as a PGM depth image instead of opening a window.
//
// SURVEY.md 8(f) rows 1 and 2 ("next" rows): CLI parity and the Wavefront OBJ reader of include/hagrid/load_obj.h (the
// reference's ObjLoader interface and behaviour, pinned to its load_obj.cpp by tests/test_obj_loader.py), so that logs of
// the two binaries can be compared line by line.
// Extension: a model name of the form soup:N generates the synthetic triangle soup of BASELINE.md instead of reading a file.
// Extension: --gpus N runs one process per GPU (this program re-executes itself N times): rank 0 builds, the grid is broadcast
// once with RCCL (ncclBroadcast from C++, include/hagrid/multi_gpu.h), every rank traces its contiguous share of the rays, the
// report is the reference's with whole-job figures.  --save-grid / --load-grid write / read the grid blob.
//
//   g++ -std=c++11 -O2 -DHOST= -DDEVICE= -Iinclude tools/hagrid_cli.cpp -o hagrid_cli -Lhagrid_amd -lhagrid_amd -lamdhip64
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include "hagrid/build.h"
#include "hagrid/load_obj.h"
#include "hagrid/mem_manager.h"
#include "hagrid/multi_gpu.h"
#include "hagrid/traverse.h"

#include "cli_report.h"

using namespace hagrid;

namespace {

struct Options {
    std::string scene, ray_file, out_image, steps_image, save_grid, load_grid;
    int gpus = 0;
    float top_density = 0.12f, snd_density = 2.4f, alpha = 0.995f;
    int exp_iters = 3, width = 1024, height = 1024;
    float clip = 0, fov = 60;
    int build_iter = 1, build_warmup = 0, bench_iter = 1, bench_warmup = 0;
    float tmin = 0, tmax = FLT_MAX;
    bool keep_alive = false, compress = false, help = false, any_hit = false;
};

enum Kind { FLAG, INT, FLOAT, STRING };
struct OptDesc { const char* s; const char* l; Kind kind; void* dst; const char* text; const char* section; };   // section: a heading printed before this option (usage)

bool parse(int argc, char** argv, Options& o, std::vector<OptDesc>& table) {
    table = {
        {"-h", "--help", FLAG, &o.help, "Shows this message"},
        {"-sx", "--width", INT, &o.width, "Sets the viewport width"},
        {"-sy", "--height", INT, &o.height, "Sets the viewport height"},
        {"-c", "--clip", FLOAT, &o.clip, "Sets the clipping distance"},
        {"-f", "--fov", FLOAT, &o.fov, "Sets the field of view"},
        {"-td", "--top-density", FLOAT, &o.top_density, "Sets the top-level density", " Construction parameters:"},
        {"-sd", "--snd-density", FLOAT, &o.snd_density, "Sets the second-level density"},
        {"-a", "--alpha", FLOAT, &o.alpha, "Sets the cell merging threshold"},
        {"-e", "--expansion", INT, &o.exp_iters, "Sets the number of expansion iterations"},
        {"-nb", "--build-iter", INT, &o.build_iter, "Sets the number of build iterations"},
        {"-wb", "--build-warmup", INT, &o.build_warmup, "Sets the number of warmup build iterations"},
        {"-k", "--keep-alive", FLAG, &o.keep_alive, "Keep the buffers alive during construction"},
        {"-z", "--compress", FLAG, &o.compress, "Compress the cells after construction"},
        {"-r", "--ray-file", STRING, &o.ray_file, "Loads rays from a file and enters benchmark mode", " Benchmarking:"},
        {"-tmin", "--tmin", FLOAT, &o.tmin, "Sets the minimum distance along every ray"},
        {"-tmax", "--tmax", FLOAT, &o.tmax, "Sets the maximum distance along every ray"},
        {"-n", "--bench-iter", INT, &o.bench_iter, "Sets the number of benchmarking iterations"},
        {"-w", "--bench-warmup", INT, &o.bench_warmup, "Sets the number of benchmarking warmup iterations"},
        {"-o", "--out", STRING, &o.out_image, "(extension) writes the traced frame as a PGM depth image", " Extensions of this front-end:"},
        {"-s", "--steps-image", STRING, &o.steps_image, "(extension) writes the per-pixel traversal step count as a PGM heat map"},
        {"-ah", "--any-hit", FLAG, &o.any_hit, "(extension) occlusion rays: a ray stops at its first intersection"},
        {"-g", "--gpus", INT, &o.gpus, "(extension) one process per GPU: the grid is built once and broadcast (RCCL), the rays are sharded"},
        {"-sg", "--save-grid", STRING, &o.save_grid, "(extension) writes the finished grid (and the triangles) to a file"},
        {"-lg", "--load-grid", STRING, &o.load_grid, "(extension) reads grid and triangles from a file instead of building"},
    };
    bool have_scene = false;
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        if (a[0] != '-') {
            if (have_scene) { std::cerr << "Cannot accept more than one model on the command line" << std::endl; return false; }
            o.scene = a; have_scene = true;
            continue;
        }
        const OptDesc* d = nullptr;
        for (const auto& t : table) if (!strcmp(a, t.s) || !strcmp(a, t.l)) d = &t;
        if (!d) { std::cerr << "Unknown argument: " << a << std::endl; return false; }
        if (d->kind == FLAG) { *static_cast<bool*>(d->dst) = true; continue; }
        if (i >= argc - 1 || (argv[i + 1][0] == '-' && d->kind == STRING)) { std::cerr << "Argument missing for: " << a << std::endl; return false; }
        const char* v = argv[++i];
        if (d->kind == INT) *static_cast<int*>(d->dst) = int(strtol(v, nullptr, 10));
        else if (d->kind == FLOAT) *static_cast<float*>(d->dst) = strtof(v, nullptr);
        else *static_cast<std::string*>(d->dst) = v;
    }
    if (!have_scene && !o.help && o.load_grid.empty()) { std::cerr << "No model specified" << std::endl; return false; }
    return true;
}

void usage(const std::vector<OptDesc>& table) {
    std::cout << "Usage: hagrid [options] file\nOptions:\n";
    for (const auto& t : table) {                      // the reference's text (main.cpp:373-396), then the extensions
        if (t.section) std::cout << t.section << "\n";
        char line[256];
        snprintf(line, sizeof(line), "  %-7s %-15s %s\n", t.s, t.l, t.text);
        std::cout << line;
    }
    std::cout << std::endl;
}

Tri make_tri(const vec3& v0, const vec3& v1, const vec3& v2) {     // packing of main.cpp:259-267
    const vec3 e1 = v0 - v1, e2 = v2 - v0, n = cross(e1, e2);
    return Tri(v0, n.x, e1, n.y, e2, n.z);
}

uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
float uniform01(uint64_t seed, uint64_t i) { return float(mix64(seed + (i + 1) * 0x9E3779B97F4A7C15ull) >> 40) * (1.0f / 16777216.0f); }

void make_soup(int n, std::vector<Tri>& tris) {     // hagrid_amd/scene.py:make_soup, same bits
    const uint64_t seed = 0x48414752494400ull + uint64_t(n);
    const float s = float(std::pow(double(n), -1.0 / 3.0));
    tris.resize(size_t(n));
    for (int i = 0; i < n; i++) {
        float u[9];
        for (int j = 0; j < 9; j++) u[j] = uniform01(seed, uint64_t(i) * 9 + j);
        const vec3 c(u[0], u[1], u[2]);
        const vec3 a = (2.0f * vec3(u[3], u[4], u[5]) - vec3(1.0f)) * s, b = (2.0f * vec3(u[6], u[7], u[8]) - vec3(1.0f)) * s;
        tris[i] = make_tri(c, c + a, c + b);
    }
}

bool load_rays(const std::string& name, std::vector<Ray>& rays, float tmin, float tmax) {   // main.cpp:277-300 format
    std::ifstream in(name, std::ifstream::binary);
    if (!in) return false;
    in.seekg(0, std::ifstream::end);
    const size_t count = size_t(in.tellg()) / (sizeof(float) * 6);
    in.seekg(0);
    std::vector<float> raw(count * 6);
    in.read(reinterpret_cast<char*>(raw.data()), std::streamsize(raw.size() * sizeof(float)));
    rays.resize(count);
    for (size_t i = 0; i < count; i++)
        rays[i] = Ray(vec3(raw[6 * i], raw[6 * i + 1], raw[6 * i + 2]), tmin, vec3(raw[6 * i + 3], raw[6 * i + 4], raw[6 * i + 5]), tmax);
    return true;
}

// SIGALRM during teardown: async-signal-safe calls only
constexpr unsigned kTeardownSeconds = 30;
void teardown_overdue(int) {
    static const char msg[] = "hagrid_cli: device teardown did not finish within 30 s; the results above are complete, ending the process\n";
    ssize_t r = write(2, msg, sizeof(msg) - 1); (void)r;
    _exit(0);
}
void arm_teardown_watchdog() {
    signal(SIGALRM, teardown_overdue);
    alarm(kTeardownSeconds);
}

extern "C" int hipSetDevice(int);      // the one HIP runtime call of this front-end (the process links libamdhip64 for the library's sake)
int hipSetDeviceShim(int device) { return hipSetDevice(device); }

// ---- multi-GPU: RCCL bound at run time, one process per GPU ---------------------------------------------------------------
struct NcclId { char internal[128]; };                       // ncclUniqueId (rccl.h:43)
struct Rccl {
    int (*get_unique_id)(NcclId*) = nullptr;
    int (*comm_init_rank)(void**, int, NcclId, int) = nullptr;
    int (*all_reduce)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    bool load() {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return false;
        get_unique_id = reinterpret_cast<int (*)(NcclId*)>(dlsym(h, "ncclGetUniqueId"));
        comm_init_rank = reinterpret_cast<int (*)(void**, int, NcclId, int)>(dlsym(h, "ncclCommInitRank"));
        all_reduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, void*)>(dlsym(h, "ncclAllReduce"));
        comm_destroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
        return get_unique_id && comm_init_rank && all_reduce && comm_destroy;
    }
};
constexpr int kNcclSum = 0, kNcclMax = 2, kNcclInt32 = 2, kNcclFloat32 = 7;       // rccl.h:448-466

// The launcher: creates the RCCL id, starts one copy of this program per GPU (rank and id travel in the environment) and waits.
int launch_ranks(char** argv, int world) {
    // All ranks are children of this process on this node: RCCL's bootstrap (the socket ncclGetUniqueId opens here, the ring the ranks form)
    // needs no other interface than the loopback, and whatever else the box has need not route to itself.  A caller's own choice stands.
    setenv("NCCL_SOCKET_IFNAME", "lo", 0);
    Rccl r;
    if (!r.load()) { std::cerr << "--gpus needs RCCL (librccl.so)" << std::endl; return 1; }
    NcclId id;
    if (r.get_unique_id(&id) != 0) { std::cerr << "ncclGetUniqueId failed" << std::endl; return 1; }
    std::string hex;
    for (unsigned char ch : id.internal) { char b[3]; snprintf(b, sizeof(b), "%02x", ch); hex += b; }
    std::vector<pid_t> kids;
    for (int rank = 0; rank < world; rank++) {
        const pid_t pid = fork();
        if (pid < 0) { std::cerr << "fork failed" << std::endl; return 1; }
        if (pid == 0) {
            setenv("HAGRID_CLI_RANK", std::to_string(rank).c_str(), 1);
            setenv("HAGRID_CLI_NCCL_ID", hex.c_str(), 1);
            // one GPU per rank; HAGRID_CLI_DEVICES=n folds the ranks onto n devices (a test of the N > 1 logic on a smaller box --
            // RCCL itself refuses two ranks on one device unless it is told otherwise)
            const char* fold = getenv("HAGRID_CLI_DEVICES");
            setenv("HAGRID_DEVICE", std::to_string(fold && atoi(fold) > 0 ? rank % atoi(fold) : rank).c_str(), 1);
            execv("/proc/self/exe", argv);
            _exit(127);
        }
        kids.push_back(pid);
    }
    int worst = 0;
    for (pid_t pid : kids) {
        int status = 0;
        waitpid(pid, &status, 0);
        const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 128;
        worst = std::max(worst, code);
    }
    return worst;
}

} // namespace

int main(int argc, char** argv) {
    Options opts;
    std::vector<OptDesc> table;
    if (argc < 2) { parse(1, argv, opts, table); usage(table); return 1; }
    if (!parse(argc, argv, opts, table)) return 1;
    if (opts.help) { usage(table); return 0; }

    // --gpus N: the first invocation only starts the ranks
    const char* env_rank = getenv("HAGRID_CLI_RANK");
    if (opts.gpus > 0 && !env_rank) return launch_ranks(argv, opts.gpus);
    const int world = opts.gpus > 0 ? opts.gpus : 1, rank = env_rank ? atoi(env_rank) : 0;
    const bool root = rank == 0;
    Rccl rccl;
    void* comm = nullptr;
    if (opts.gpus > 0) {
        const char* hex = getenv("HAGRID_CLI_NCCL_ID");
        NcclId id;
        if (!rccl.load() || !hex || strlen(hex) != 256) { std::cerr << "rank " << rank << ": no RCCL / no id" << std::endl; return 1; }
        for (int i = 0; i < 128; i++) { unsigned v = 0; sscanf(hex + 2 * i, "%2x", &v); id.internal[i] = char(v); }
        if (hipSetDeviceShim(atoi(getenv("HAGRID_DEVICE") ? getenv("HAGRID_DEVICE") : "0")) != 0 || rccl.comm_init_rank(&comm, world, id, rank) != 0) { std::cerr << "rank " << rank << ": ncclCommInitRank failed" << std::endl; return 1; }
    }

    std::vector<Tri> host_tris;
    MemManager mem(opts.keep_alive);
    Tri* tris = nullptr;
    int num_tris = 0;
    Grid grid;
    grid.entries = nullptr; grid.cells = nullptr; grid.ref_ids = nullptr; grid.small_cells = nullptr;
    const bool build_here = root && opts.load_grid.empty();
    if (build_here) {
        if (opts.scene.compare(0, 5, "soup:") == 0) make_soup(atoi(opts.scene.c_str() + 5), host_tris);
        else if (!load_obj_triangles(opts.scene, host_tris)) {
            std::cerr << "Scene cannot be loaded (file not present or contains errors)" << std::endl;
            return 1;
        }
        hagrid_cli::report_scene(std::cout, host_tris.size());
        num_tris = int(host_tris.size());
        tris = mem.alloc<Tri>(host_tris.size());
        mem.copy<Copy::HST_TO_DEV>(tris, host_tris.data(), host_tris.size());
    } else if (root) {
        if (!load_grid(mem, opts.load_grid, grid, tris, num_tris)) { std::cerr << "Grid file cannot be loaded" << std::endl; return 1; }
        hagrid_cli::report_scene(std::cout, size_t(num_tris));
    }

    auto construct = [&] {
        build_grid(mem, tris, num_tris, grid, opts.top_density, opts.snd_density);
        merge_grid(mem, grid, opts.alpha);
        flatten_grid(mem, grid);
        expand_grid(mem, grid, tris, opts.exp_iters);
        if (opts.compress) compress_grid(mem, grid);
    };
    auto release = [&] {
        mem.free(grid.entries); mem.free(grid.cells); mem.free(grid.ref_ids); mem.free(grid.small_cells);
        grid.entries = nullptr; grid.cells = nullptr; grid.ref_ids = nullptr; grid.small_cells = nullptr;
    };
    double total_time = 0;
    if (build_here) {
        for (int i = 0; i < opts.build_warmup; i++) { release(); construct(); }
        for (int i = 0; i < opts.build_iter; i++) { release(); total_time += profile(construct); }
        if (opts.compress && !grid.small_cells) std::cerr << "Could not compress grid. Continuing with uncompressed structure." << std::endl;
        if (!opts.save_grid.empty()) save_grid(mem, grid, tris, num_tris, opts.save_grid);
    }
    if (comm) {                                      // the ONE exchange step: the finished grid from rank 0 to everybody
        const double ms = profile([&] { broadcast_grid(mem, grid, tris, num_tris, comm, rank, 0); });
        if (root) std::cout << world << " rank(s), grid broadcast in " << ms << " ms" << std::endl;
    }

    if (root) {
        const ivec3 dims = grid.dims << grid.shift;
        hagrid_cli::report_grid(std::cout, build_here ? total_time / opts.build_iter : -1.0, dims.x, dims.y, dims.z, grid.num_cells, grid.num_refs);
        hagrid_cli::report_memory(std::cout, size_t(grid.num_cells) * (grid.small_cells ? sizeof(SmallCell) : sizeof(Cell)), size_t(grid.num_entries) * sizeof(int),
                                  size_t(grid.num_refs) * sizeof(int), size_t(num_tris) * sizeof(Tri), mem.max_usage());
    }

    setup_traversal(grid);
    const float scene_size = length(grid.bbox.extents());
    const vec3 center = grid.bbox.center();
    if (opts.clip <= 0) opts.clip = scene_size;

    std::vector<Ray> host_rays;
    if (!opts.ray_file.empty()) {
        if (root) std::cout << "Entering benchmark mode" << std::endl;
        if (!load_rays(opts.ray_file, host_rays, opts.tmin, opts.tmax)) { std::cerr << "Cannot load ray file" << std::endl; return 1; }
    } else {
        // one frame of the viewer's initial view (main.cpp:572-579, :592-598): eye at the scene centre, looking down +z
        if (root) std::cout << "Tracing one " << opts.width << "x" << opts.height << " frame (no interactive viewer in this front-end)" << std::endl;
        const vec3 eye = center, forward(0.0f, 0.0f, 1.0f), up(0.0f, 1.0f, 0.0f);
        const float f = tanf(float(M_PI) * opts.fov / 360.0f), ratio = float(opts.width) / float(opts.height);
        const vec3 dir = normalize((eye + forward * 100.0f) - eye), right = normalize(cross(dir, up)) * (f * ratio), cup = normalize(cross(right, dir)) * f;
        host_rays.resize(size_t(opts.width) * opts.height);
        for (int y = 0; y < opts.height; y++)
            for (int x = 0; x < opts.width; x++) {
                const float kx = 2 * x / float(opts.width) - 1, ky = 1 - 2 * y / float(opts.height);
                host_rays[size_t(y) * opts.width + x] = Ray(eye, 0.0f, dir + right * kx + cup * ky, opts.clip);
            }
    }
    // this rank's contiguous share of the batch (the whole batch without --gpus)
    const size_t all_rays = host_rays.size();
    size_t first = 0, last = all_rays;
    shard_range(all_rays, rank, world, first, last);
    if (world > 1) host_rays = std::vector<Ray>(host_rays.begin() + first, host_rays.begin() + last);
    Ray* rays = mem.alloc<Ray>(host_rays.size());
    Hit* hits = mem.alloc<Hit>(host_rays.size());
    mem.copy<Copy::HST_TO_DEV>(rays, host_rays.data(), host_rays.size());
    auto trace = [&] {
        if (opts.any_hit) traverse_grid_any_hit(grid, tris, rays, hits, int(host_rays.size()));
        else              traverse_grid(grid, tris, rays, hits, int(host_rays.size()));
    };
    for (int i = 0; i < opts.bench_warmup; i++) trace();
    std::vector<float> timings;
    for (int i = 0; i < std::max(opts.bench_iter, 1); i++) timings.push_back(profile(trace));
    std::vector<Hit> host_hits(host_rays.size());
    mem.copy<Copy::DEV_TO_HST>(host_hits.data(), hits, host_hits.size());
    int intr = 0;
    for (const auto& h : host_hits) intr += h.id >= 0;
    if (comm) {
        // whole-job figures: an iteration takes as long as its slowest rank, intersections add up
        float* d_t = mem.alloc<float>(timings.size());
        int* d_n = mem.alloc<int>(1);
        mem.copy<Copy::HST_TO_DEV>(d_t, timings.data(), timings.size());
        mem.copy<Copy::HST_TO_DEV>(d_n, &intr, 1);
        if (rccl.all_reduce(d_t, d_t, timings.size(), kNcclFloat32, kNcclMax, comm, nullptr) != 0 ||
            rccl.all_reduce(d_n, d_n, 1, kNcclInt32, kNcclSum, comm, nullptr) != 0) { std::cerr << "ncclAllReduce failed" << std::endl; return 1; }
        mem.copy<Copy::DEV_TO_HST>(timings.data(), d_t, timings.size());
        mem.copy<Copy::DEV_TO_HST>(&intr, d_n, 1);
        mem.free(d_t); mem.free(d_n);
    }
    if (root) hagrid_cli::report_timings(std::cout, std::vector<double>(timings.begin(), timings.end()), all_rays, intr);
    if (!opts.out_image.empty() && opts.ray_file.empty() && world == 1) {
        std::ofstream img(opts.out_image, std::ofstream::binary);
        img << "P5\n" << opts.width << " " << opts.height << "\n255\n";
        for (const auto& h : host_hits) img.put(char(h.id >= 0 ? std::min(255.0f, 255.0f * h.t / opts.clip) : 255));
    }
    if (!opts.steps_image.empty() && opts.ray_file.empty() && world == 1) {
        // the picture the reference's viewer shows: its kernel returns the step count in Hit::id (traverse.cu:80,93) and
        // main.cpp:100-107 maps it to a colour; here the count comes from the statistics entry point
        int* steps = mem.alloc<int>(host_rays.size());
        hagrid_grid pod = detail::to_pod(grid);
        detail::check(detail::current_ctx(), hagrid_traverse_grid_stats(detail::current_ctx(), &pod, tris, rays, hits, int(host_rays.size()), steps, nullptr));
        std::vector<int> host_steps(host_rays.size());
        mem.copy<Copy::DEV_TO_HST>(host_steps.data(), steps, host_steps.size());
        const int top = std::max(1, *std::max_element(host_steps.begin(), host_steps.end()));
        std::ofstream img(opts.steps_image, std::ofstream::binary);
        img << "P5\n" << opts.width << " " << opts.height << "\n255\n";
        for (int v : host_steps) img.put(char(255 * v / top));
        std::cout << "Steps per ray: max " << top << ", mean " << std::accumulate(host_steps.begin(), host_steps.end(), 0.0) / host_steps.size() << std::endl;
        mem.free(steps);
    }
    // Everything the caller asked for is on stdout and in its files (the image outputs above are work, not teardown: they may take as
    // long as they take).  What follows returns device memory and takes the runtime down; a runtime that does not come back from that
    // (seen once in the round-2 driver run: the report printed, the process never exited) must not hold the caller: after
    // kTeardownSeconds the process says so on stderr and ends with the status of its work.
    std::cout.flush();
    arm_teardown_watchdog();
    mem.free(rays); mem.free(hits); release(); mem.free(tris);
    if (comm) rccl.comm_destroy(comm);
    return 0;
}
