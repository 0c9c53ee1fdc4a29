// Headless stand-in for the reference's GLFW shell (src/main.cpp:117-159): where the reference opens a
// 1000x1000 window and ray-traces on key `T` (src/main.cpp:9-10,69-70), this program loads the scene,
// calls Flyscene::raytraceScene() once and exits.
//
//   raytrace [--obj PATH] [--size WxH] [--out FILE.ppm] [--max-reflections N] [--ssaa N]
//            [--hard-shadows-only] [--light X,Y,Z]... [--move DX,DY,DZ] [--device-builder]
//            [--frames N [--in-flight F]]
//
// --frames N renders N more views of a slow orbit straight into device buffers, F of them enqueued at a time, each
// on a HIP stream of its own (rtamd_render_device keeps one workspace per stream): the way a host that renders a
// sequence gets the frames to overlap on the GPU (F up to 16; above 4 the program asks the HIP runtime for sixteen
// hardware queues, unless GPU_MAX_HW_QUEUES is already set).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtamd_flyscene.h"
#include "flyscene_host.hpp"

int main(int argc, char **argv) {
    rtamd::Flyscene fly;
    int width = 1000, height = 1000; // src/main.cpp:9-10
    int max_reflections = -1, ssaa = -1, frames = 0, in_flight = 4;
    float move[3] = {0, 0, 0};
    bool moved = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--obj") {
            fly.model_path = next();
        } else if (a == "--size") {
            if (std::sscanf(next(), "%dx%d", &width, &height) != 2) return std::fprintf(stderr, "--size WxH\n"), 2;
        } else if (a == "--out") {
            fly.output_path = next();
        } else if (a == "--max-reflections") {
            max_reflections = std::atoi(next());
        } else if (a == "--ssaa") {
            ssaa = std::atoi(next());
        } else if (a == "--hard-shadows-only") {
            fly.hard_shadow = true, fly.soft_points = 0;
        } else if (a == "--light") {
            rtamd::Light l{{0, 0, 0}, {1, 1, 1}};
            if (std::sscanf(next(), "%f,%f,%f", &l.position[0], &l.position[1], &l.position[2]) != 3) return std::fprintf(stderr, "--light X,Y,Z\n"), 2;
            fly.lights.push_back(l);
            fly.lights_defaulted = true; // explicit lights replace the built-in one
        } else if (a == "--frames") {
            frames = std::atoi(next());
        } else if (a == "--in-flight") {
            in_flight = std::min(std::max(std::atoi(next()), 1), 16);
        } else if (a == "--device-builder") {
            fly.build.builder = RTAMD_BUILDER_DEVICE_LBVH;
        } else if (a == "--move") {
            if (std::sscanf(next(), "%f,%f,%f", &move[0], &move[1], &move[2]) != 3) return std::fprintf(stderr, "--move DX,DY,DZ\n"), 2;
            moved = true;
        } else {
            std::fprintf(stderr, "unknown option %s\n", a.c_str());
            return 2;
        }
    }
    // Frames on different streams overlap only while each stream has a hardware queue of its own; the HIP runtime
    // gives a process four unless told otherwise before its first call.
    if (frames > 0 && in_flight > 4) setenv("GPU_MAX_HW_QUEUES", "16", 0);
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::printf("Initializing scene ...\n");
    const auto t0 = clk::now();
    int rc = fly.initialize(width, height);
    if (rc != RTAMD_OK) {
        std::fprintf(stderr, "initialize failed (%d): %s\n", rc, rtamd_last_error());
        return 1;
    }
    if (max_reflections >= 0) fly.params.max_reflections = max_reflections;
    if (ssaa > 0) fly.params.ssaa_level = ssaa;
    if (moved) {
        fly.getCamera()->translate(move[0], move[1], move[2]);
        fly.getCamera()->updateViewMatrix();
    }
    const auto t1 = clk::now();
    rtamd_scene_info info;
    rtamd_scene_get_info(fly.scene(), &info);
    std::printf("Scene: %d faces, %d nodes, depth %d, %.1f MB on device %d; init %.0fms (BVH %.0fms)\n", info.num_faces, info.num_nodes, info.depth,
                info.device_bytes / 1e6, info.device, ms(t0, t1), info.build_ms);
    std::printf("-------------------------\nStarting ray tracing ...\n");
    rc = fly.raytraceScene();
    const auto t2 = clk::now();
    if (rc != RTAMD_OK) {
        std::fprintf(stderr, "raytraceScene failed (%d): %s\n", rc, rtamd_last_error());
        return 1;
    }
    std::printf("Done! Total time: %.0fms -> %s\n", ms(t1, t2), fly.output_path.c_str());
    if (frames > 0) {
        // a sequence: frame k goes to stream k mod F and buffer k mod F, nothing comes back to the host in between
        std::vector<hipStream_t> streams((size_t)in_flight);
        std::vector<float *> rgb((size_t)in_flight, nullptr);
        const size_t bytes = (size_t)width * height * 3 * sizeof(float);
        for (int k = 0; k < in_flight; ++k)
            if (hipStreamCreateWithFlags(&streams[(size_t)k], hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&rgb[(size_t)k], bytes) != hipSuccess)
                return std::fprintf(stderr, "HIP stream or buffer allocation failed\n"), 1;
        rtamd_stats stats;
        uint64_t rays = 0;
        const auto t3 = clk::now();
        for (int k = 0; k < frames; ++k) {
            fly.getCamera()->rotate(0.0f, 0.002f); // as dragging the mouse does in the reference's window
            fly.getCamera()->updateViewMatrix();
            rtamd_camera cam;
            fly.getCamera()->fill(cam);
            rc = rtamd_render_device(fly.scene(), &cam, &fly.params, rgb[(size_t)(k % in_flight)], nullptr, nullptr, streams[(size_t)(k % in_flight)]);
            if (rc != RTAMD_OK) return std::fprintf(stderr, "frame %d failed (%d): %s\n", k, rc, rtamd_last_error()), 1;
        }
        if (rtamd_render_stats(fly.scene(), &stats) != RTAMD_OK) return std::fprintf(stderr, "%s\n", rtamd_last_error()), 1; // waits for all frames
        const auto t4 = clk::now();
        rays = stats.rays_primary + stats.rays_shadow + stats.rays_bounce;
        std::printf("%d frames, %d in flight: %.3f ms per frame, %.0f Mrays/s (last frame: %llu rays)\n", frames, in_flight, ms(t3, t4) / frames,
                    (double)rays * frames / ms(t3, t4) / 1e3, (unsigned long long)rays);
        for (int k = 0; k < in_flight; ++k) {
            (void)hipFree(rgb[(size_t)k]);
            (void)hipStreamDestroy(streams[(size_t)k]);
        }
    }
    return 0;
}
