// nbody_headless.cpp — the reference's headless run (TestProject/main.cpp:231-368, the
// enableVisualization == false branch) as a non-interactive host program on top of the C-ABI.
//
//   main.cpp step                                   here
//   cudaMallocHost x3, fill_with_random4/zeroes4    nbody_malloc_host, --init libc|ref|plummer   (:250-272)
//   cudaMalloc + cudaMemcpy H2D x3                  nbody_malloc_device, nbody_memcpy_h2d         (:275-283,352-354)
//   askForKernelType / Visualization / StepsNumber  argv; --interactive reads the same three answers (:163-228)
//   simulationLoopNoVisual                          `steps` x simulate() (--sync-each-step) or one queued nbody_step (:142-160)
//   (results discarded, no timing)                  D2H, --dump state, timing, one JSON line
//   cudaFree / cudaFreeHost                         nbody_free_*                                   (:358-366)
//   device 0 only (kernel.cu:630, main.cpp:287)     --ngpu G: G ranks in this process (one thread and one GPU each) over
//                                                   nbody_shard_* + nbody_comm_rccl_* (RCCL all-gather / exchange), or
//                                                   --transport local: nbody_comm_local_* (hipMemcpyPeerAsync pulls between the
//                                                   rank threads, no RCCL; --share-devices: more ranks than GPUs, a rehearsal)
//   fp32 only                                       --precision f64: the build's double variant (nbody_step_f64, 1 GPU)
//
// State files (--dump P / --load P): P.json {n, steps_done, dt, eps2} + P.x.f4 / P.v.f4 / P.a.f4,
// raw little-endian float4[N] — exactly the three arrays main.cpp owns (main.cpp:232-241).
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "nbody.h"

struct float4 { float x, y, z, w; };
#include "nbody_compat.hpp"

static void die(const std::string& m)
{
    std::cerr << m << std::endl;
    std::exit(EXIT_FAILURE);
}
static void ok(int rc)
{
    if (rc != NBODY_OK) die(std::string("nbody: ") + nbody_last_error());
}

static bool write_file(const std::string& path, const void* p, size_t bytes)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool good = std::fwrite(p, 1, bytes, f) == bytes;
    std::fclose(f);
    return good;
}
static bool read_file(const std::string& path, void* p, size_t bytes)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool good = std::fread(p, 1, bytes, f) == bytes;
    std::fclose(f);
    return good;
}

int main(int argc, char** argv)
{
    int n = N_BODIES, steps = 10, kernel = NBODY_KERNEL_FAST, sync_each = 0, interactive = 0, json = 1, ngpu = 1, timeout_s = 900, autotune = 0, equal_mass = -1, inplace_sums = -1, clock = 0;
    std::string transport = "rccl";   // --ngpu: rccl (librccl, one GPU per rank) | local (hipMemcpyPeerAsync pulls between the rank threads, no RCCL)
    bool share_devices = false;
    bool f64 = false, force_shard = false;
    long steps_done = 0;
    double force_ms = 0.0;
    int force_launches = 0;
    nbody_clock_report clk{};
    float dt = DT, eps2 = EPS2;
    unsigned long long seed = 12345;
    std::string init = "libc", dump, load;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) die("missing value for " + a); return argv[++i]; };
        if (a == "--n") n = std::atoi(val());
        else if (a == "--steps") steps = std::atoi(val());
        else if (a == "--dt") dt = (float)std::atof(val());
        else if (a == "--eps2") eps2 = (float)std::atof(val());
        else if (a == "--seed") seed = std::strtoull(val(), nullptr, 10);
        else if (a == "--init") init = val();            // libc (utils.cpp:30-37) | ref | plummer
        else if (a == "--kernel") {
            std::string k = val();
            kernel = (k == "strict" || k == "1") ? NBODY_KERNEL_STRICT : k == "onesided" ? NBODY_KERNEL_ONESIDED
                     : k == "symmetric" ? NBODY_KERNEL_SYMMETRIC : NBODY_KERNEL_FAST;
        }
        else if (a == "--ngpu") ngpu = std::atoi(val());
        else if (a == "--transport") { transport = val(); if (transport != "rccl" && transport != "local") die("--transport rccl|local"); }
        else if (a == "--share-devices") share_devices = true;   // --transport local only: more ranks than GPUs, rank r on device r % ndev (rehearsal)
        else if (a == "--timeout") timeout_s = std::atoi(val());   // --ngpu: seconds before a stalled collective is given up (0 = wait for ever)
        else if (a == "--shard") force_shard = true;     // run through nbody_shard_* + RCCL even with one GPU
        else if (a == "--precision") { std::string q = val(); if (q != "f32" && q != "f64") die("--precision f32|f64"); f64 = q == "f64"; }
        else if (a == "--dump") dump = val();
        else if (a == "--load") load = val();
        else if (a == "--sync-each-step") sync_each = 1;
        else if (a == "--no-equal-mass") equal_mass = 0;  // nbody_ctx_set_equal_mass(0): the general pair arithmetic whatever the masses
        else if (a == "--inplace-sums") { std::string q = val(); if (q != "auto" && q != "on" && q != "off") die("--inplace-sums auto|on|off"); inplace_sums = q == "on" ? 1 : q == "off" ? 0 : -1; }   // nbody_ctx_set_inplace_sums: block sums added in place (no slab workspace)
        else if (a == "--clock") clock = 1;              // nbody_ctx_timing(ctx, 2): the force launches' event time, shader cycles and the shader clock under load in the JSON line (queued single-GPU stepping)
        else if (a == "--autotune") autotune = 1;        // measure the decompositions on this device first (single GPU, fast kernel)
        else if (a == "--interactive") interactive = 1;
        else if (a == "--quiet") json = 0;
        else die("unknown option " + a + "\nusage: nbody_headless [--n N] [--steps K] [--dt f] [--eps2 f] [--init libc|ref|plummer] [--seed S]"
                 " [--kernel fast|strict|onesided|symmetric] [--autotune] [--no-equal-mass] [--ngpu G] [--transport rccl|local] [--share-devices] [--timeout S] [--shard] [--precision f32|f64] [--dump P] [--load P] [--sync-each-step]"
                 " [--interactive]");
    }
    if (interactive) {
        // the reference's three prompts (main.cpp:163-228); only the headless all-pairs answers run here
        std::string s;
        std::cout << "Choose the kernel (0 = basic all-pairs, 1 = reduction): ";
        std::getline(std::cin, s);
        if (s != "0") die("only the basic all-pairs kernel (0) is built; the reduction kernels need the visual path");
        std::cout << "Enable visualization? (y/n): ";
        std::getline(std::cin, s);
        if (s != "n" && s != "N") die("visualization is out of scope of this build; answer n");
        std::cout << "Number of steps: ";
        std::getline(std::cin, s);
        steps = std::atoi(s.c_str());
    }
    if (n < 0 || steps < 0) die("n and steps must be >= 0");
    if (ngpu < 1 || ngpu > NBODY_MAX_RANKS) die("--ngpu must be in [1, 64]");
    if (f64 && (ngpu > 1 || sync_each || force_shard)) die("--precision f64 is a single-GPU variant stepped through nbody_step_f64");
    if (ngpu > 1 && sync_each) die("--sync-each-step is the reference's single-device loop");
    const size_t bytes = sizeof(float4) * (size_t)n;

    float4 *bodies = nullptr, *velocity = nullptr, *accelerations = nullptr;
    ok(nbody_malloc_host((void**)&bodies, bytes));
    ok(nbody_malloc_host((void**)&velocity, bytes));
    ok(nbody_malloc_host((void**)&accelerations, bytes));
    if (!load.empty()) {
        if (!read_file(load + ".x.f4", bodies, bytes) || !read_file(load + ".v.f4", velocity, bytes))
            die("cannot read state " + load + ".{x,v}.f4 for n=" + std::to_string(n));
        fill_with_zeroes4(accelerations, n);
        if (FILE* f = std::fopen((load + ".json").c_str(), "r")) {
            char buf[512] = {0};
            if (std::fread(buf, 1, sizeof buf - 1, f) > 0)
                if (const char* p = std::strstr(buf, "\"steps_done\":")) steps_done = std::atol(p + 13);
            std::fclose(f);
        }
    } else {
        if (init == "libc") fill_with_random4(bodies, n);
        else if (init == "ref") ok(nbody_fill_seeded((nbody_float4*)bodies, n, 0, seed));
        else if (init == "plummer") ok(nbody_fill_seeded((nbody_float4*)bodies, n, 1, seed));
        else die("unknown --init " + init);
        fill_with_zeroes4(velocity, n);
        fill_with_zeroes4(accelerations, n);
    }

    const char* kname = kernel == NBODY_KERNEL_STRICT ? "strict" : kernel == NBODY_KERNEL_ONESIDED ? "onesided"
                        : kernel == NBODY_KERNEL_SYMMETRIC ? "symmetric" : "fast";
    double secs = 0.0;
    unsigned long long inplace_fallback_waves = 0;
    int autotuned_choice = -1;
    if (ngpu > 1 || force_shard) {
        // G ranks in this process: one thread, one device, one context, one RCCL communicator and one shard each
        int ndev = 0;
        ok(nbody_device_count(&ndev));
        const bool local = transport == "local";
        if (share_devices && !local) die("--share-devices needs --transport local (RCCL refuses two ranks on one device)");
        if (ngpu > ndev && !share_devices) die("--ngpu " + std::to_string(ngpu) + " but only " + std::to_string(ndev) + " device(s) visible");
        if (ndev < 1) die("no device");
        char uid[128];
        nbody_local_group* group = nullptr;
        if (local) ok(nbody_comm_local_group_create(&group, ngpu, timeout_s > 0 ? (double)timeout_s : 0.0));
        else ok(nbody_comm_rccl_unique_id(uid));
        // Every step that can fail LOCALLY (device, context, allocations) is taken before a collective one, and the ranks
        // agree on success at a gate before entering it: a rank that failed never leaves its peers waiting inside
        // ncclCommInitRank or the first all-gather. What can still stall (a collective itself) is bounded by a deadline
        // on the main thread. Results are downloaded into per-rank buffers and merged after the join.
        struct Gate {   // reusable barrier over the rank threads carrying a shared failure flag
            std::mutex mu;
            std::condition_variable cv;
            int waiting = 0, generation = 0, parties;
            bool failed = false;
            explicit Gate(int n) : parties(n) {}
            bool pass(bool ok_here)   // returns true when EVERY rank arrived with ok_here == true (now and at every earlier gate)
            {
                std::unique_lock<std::mutex> lk(mu);
                if (!ok_here) failed = true;
                const int gen = generation;
                if (++waiting == parties) { waiting = 0; ++generation; cv.notify_all(); }
                else cv.wait(lk, [&] { return generation != gen; });
                return !failed;
            }
        } gate(ngpu);
        std::vector<std::string> errors(ngpu);
        std::vector<double> rank_secs(ngpu, 0.0);
        struct RankOut { int i0 = 0, i1 = 0; std::vector<nbody_float4> x, v, a; };
        std::vector<RankOut> outs(ngpu);
        std::mutex done_mu;
        std::condition_variable done_cv;
        int done = 0;
        std::printf("Starting the simulation...\n");
        auto rank_main = [&](int r) {
            auto note = [&](const char* what) { if (errors[r].empty()) errors[r] = std::string(what) + ": " + nbody_last_error(); };
            nbody_ctx* ctx = nullptr;
            nbody_comm comm{};
            nbody_shard* sh = nullptr;
            bool good = true;
            // 1. local: device, context
            const int dev = r % ndev;
            if (nbody_ctx_create(&ctx, dev) != NBODY_OK) { note("nbody_ctx_create"); good = false; }
            else if (nbody_ctx_set_params(ctx, dt, eps2) != NBODY_OK || nbody_ctx_set_kernel(ctx, kernel, 0, 0, 0) != NBODY_OK ||
                     nbody_ctx_set_equal_mass(ctx, equal_mass) != NBODY_OK) { note("context setup"); good = false; }
            if (gate.pass(good)) {
                // 2. collective: the communicator (every rank enters, or none does)
                if (local) { if (nbody_comm_local_create(&comm, group, r, dev) != NBODY_OK) { note("nbody_comm_local_create"); good = false; } }
                else if (nbody_comm_rccl_create(&comm, r, ngpu, uid, dev) != NBODY_OK) { note("nbody_comm_rccl_create"); good = false; }
                // 3. local: the shard's device arrays, the upload
                if (good && nbody_shard_create(&sh, ctx, r, ngpu, n, &comm) != NBODY_OK) { note("nbody_shard_create"); good = false; }
                if (good && nbody_shard_upload(sh, (const nbody_float4*)bodies) != NBODY_OK) { note("nbody_shard_upload"); good = false; }
                if (good && !load.empty() && nbody_shard_upload_velocity(sh, (const nbody_float4*)velocity) != NBODY_OK) { note("nbody_shard_upload_velocity"); good = false; }
                // every rank has read the shared host arrays before anybody steps (and nobody writes them before the join)
                if (gate.pass(good)) {
                    const auto t0 = std::chrono::steady_clock::now();
                    if (nbody_shard_step(sh, steps) != NBODY_OK || nbody_shard_sync(sh) != NBODY_OK) {
                        note("nbody_shard_step");
                        good = false;
                        if (local) nbody_comm_local_abort(group);   // peers waiting at a rendezvous give up instead of hanging
                    }
                    rank_secs[r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    nbody_shard_plan_t plan;
                    if (good && nbody_shard_get_plan(sh, &plan) == NBODY_OK) {
                        RankOut& o = outs[r];
                        o.i0 = plan.i0; o.i1 = plan.i1;
                        o.x.resize(plan.shard); o.v.resize(plan.shard); o.a.resize(plan.shard);
                        if (nbody_shard_download(sh, o.x.data(), o.v.data(), o.a.data()) != NBODY_OK) { note("nbody_shard_download"); good = false; }
                    }
                }
            }
            if (!good && errors[r].empty()) errors[r] = "skipped: another rank failed";
            if (sh) nbody_shard_destroy(sh);
            if (local) nbody_comm_local_destroy(&comm);
            else nbody_comm_rccl_destroy(&comm);
            if (ctx) nbody_ctx_destroy(ctx);
            std::lock_guard<std::mutex> lk(done_mu);
            ++done;
            done_cv.notify_all();
        };
        std::vector<std::thread> threads;
        for (int r = 0; r < ngpu; ++r) threads.emplace_back(rank_main, r);
        {
            std::unique_lock<std::mutex> lk(done_mu);
            const bool finished = timeout_s <= 0 ? (done_cv.wait(lk, [&] { return done == ngpu; }), true)
                                                 : done_cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return done == ngpu; });
            if (!finished) {   // a collective never completed: report and leave without unwinding the stuck threads
                std::fprintf(stderr, "nbody_headless: %d of %d ranks still running after %d s (a collective stalled?); giving up\n",
                             ngpu - done, ngpu, timeout_s);
                for (int r = 0; r < ngpu; ++r)
                    if (!errors[r].empty()) std::fprintf(stderr, "rank %d: %s\n", r, errors[r].c_str());
                std::fflush(stderr);
                std::_Exit(3);
            }
        }
        for (auto& t : threads) t.join();
        if (group) nbody_comm_local_group_destroy(group);
        bool any_error = false;
        for (int r = 0; r < ngpu; ++r)
            if (!errors[r].empty() && errors[r].rfind("skipped", 0) != 0) { std::cerr << "rank " << r << ": " << errors[r] << std::endl; any_error = true; }
        for (int r = 0; r < ngpu && !any_error; ++r)
            if (!errors[r].empty()) { std::cerr << "rank " << r << ": " << errors[r] << std::endl; any_error = true; }
        if (any_error) return EXIT_FAILURE;
        for (int r = 0; r < ngpu; ++r) {
            const RankOut& o = outs[r];
            for (int i = o.i0; i < o.i1 && i < n; ++i) {   // the own blocks back into the whole-system host arrays
                std::memcpy(&bodies[i], &o.x[i - o.i0], sizeof(float4));
                std::memcpy(&velocity[i], &o.v[i - o.i0], sizeof(float4));
                std::memcpy(&accelerations[i], &o.a[i - o.i0], sizeof(float4));
            }
            if (rank_secs[r] > secs) secs = rank_secs[r];
        }
        std::printf("Simulation complete\n");
    } else if (f64) {
        std::vector<nbody_double4> hx(n), hv(n), ha(n);
        for (int i = 0; i < n; ++i) {
            hx[i] = {bodies[i].x, bodies[i].y, bodies[i].z, bodies[i].w};
            hv[i] = {velocity[i].x, velocity[i].y, velocity[i].z, 0.0};
            ha[i] = {0.0, 0.0, 0.0, 0.0};
        }
        const size_t b8 = sizeof(nbody_double4) * (size_t)n;
        void *dx = nullptr, *dv = nullptr, *da = nullptr;
        ok(nbody_malloc_device(&dx, b8));
        ok(nbody_malloc_device(&dv, b8));
        ok(nbody_malloc_device(&da, b8));
        ok(nbody_memcpy_h2d(dx, hx.data(), b8));
        ok(nbody_memcpy_h2d(dv, hv.data(), b8));
        ok(nbody_memcpy_h2d(da, ha.data(), b8));
        nbody_ctx* ctx = nullptr;
        ok(nbody_default_ctx(&ctx));
        std::printf("Starting the simulation...\n");
        const auto t0 = std::chrono::steady_clock::now();
        ok(nbody_step_f64(ctx, (nbody_double4*)dx, (nbody_double4*)da, (nbody_double4*)dv, n, steps, (double)dt, (double)eps2));
        ok(nbody_ctx_sync(ctx));
        secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("Simulation complete\n");
        ok(nbody_memcpy_d2h(hx.data(), dx, b8));
        ok(nbody_memcpy_d2h(hv.data(), dv, b8));
        ok(nbody_memcpy_d2h(ha.data(), da, b8));
        if (!dump.empty() && (!write_file(dump + ".x.f8", hx.data(), b8) || !write_file(dump + ".v.f8", hv.data(), b8) ||
                              !write_file(dump + ".a.f8", ha.data(), b8)))
            die("cannot write state " + dump);
        for (int i = 0; i < n; ++i) {   // the fp32 view of the result (what the JSON line and the .f4 files carry)
            bodies[i] = {(float)hx[i].x, (float)hx[i].y, (float)hx[i].z, (float)hx[i].w};
            velocity[i] = {(float)hv[i].x, (float)hv[i].y, (float)hv[i].z, 0.0f};
            accelerations[i] = {(float)ha[i].x, (float)ha[i].y, (float)ha[i].z, 0.0f};
        }
        ok(nbody_free_device(dx));
        ok(nbody_free_device(dv));
        ok(nbody_free_device(da));
        kname = "f64";
    } else {
    float4 *d_bodies = nullptr, *d_velocity = nullptr, *d_accelerations = nullptr;
    ok(nbody_malloc_device((void**)&d_velocity, bytes));
    ok(nbody_malloc_device((void**)&d_accelerations, bytes));
    ok(nbody_malloc_device((void**)&d_bodies, bytes));
    ok(nbody_memcpy_h2d(d_velocity, velocity, bytes));
    ok(nbody_memcpy_h2d(d_accelerations, accelerations, bytes));
    ok(nbody_memcpy_h2d(d_bodies, bodies, bytes));

    nbody_ctx* ctx = nullptr;
    ok(nbody_default_ctx(&ctx));                         // the context simulate() uses
    ok(nbody_ctx_set_params(ctx, dt, eps2));
    ok(nbody_ctx_set_kernel(ctx, kernel, 0, 0, 0));
    ok(nbody_ctx_set_equal_mass(ctx, equal_mass));
    ok(nbody_ctx_set_inplace_sums(ctx, inplace_sums));
    ok(nbody_ctx_reserve(ctx, n));
    if (autotune && kernel == NBODY_KERNEL_FAST && n > 0) {
        int choice = 0;
        double us = 0.0;
        ok(nbody_ctx_autotune(ctx, (const nbody_float4*)d_bodies, n, 50, &choice, &us));
        std::printf("autotune: decomposition %d, %.2f us per step\n", choice, us);
    }

    if (sync_each) ok(nbody_simulate_prepare((const nbody_float4*)d_bodies, n));   // one-off work of simulate()'s first call, outside the timed loop
    std::printf("Starting the simulation...\n");         // main.cpp:145
    const auto t0 = std::chrono::steady_clock::now();
    if (sync_each) {
        for (int counter = 0; counter < steps; ++counter) {   // the reference's loop, one sync per step
            try {
                simulate(d_bodies, d_accelerations, d_velocity, n);
            } catch (const std::exception& e) {
                std::cerr << e.what() << std::endl;
                return EXIT_FAILURE;
            }
        }
    } else {
        if (clock) ok(nbody_ctx_timing(ctx, 2));
        ok(nbody_step(ctx, (nbody_float4*)d_bodies, (nbody_float4*)d_accelerations, (nbody_float4*)d_velocity, n, steps));
        ok(nbody_ctx_sync(ctx));
    }
    secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (clock && !sync_each) {
        ok(nbody_ctx_timing_read(ctx, &force_ms, &force_launches));
        ok(nbody_ctx_clock_read(ctx, &clk));
        ok(nbody_ctx_timing(ctx, 0));
    }
    std::printf("Simulation complete\n");                // main.cpp:158
    if (sync_each) ok(nbody_ctx_autotuned(ctx, n, &autotuned_choice, nullptr, nullptr));   // what simulate()'s one-off measurement decided for this size (-1: none)
    if (sync_each) ok(nbody_ctx_fused_inplace_stats(ctx, &inplace_fallback_waves));   // waves of the in-place step that wrote to the spare array (0 when the GPU was ours alone)

    ok(nbody_memcpy_d2h(bodies, d_bodies, bytes));
    ok(nbody_memcpy_d2h(velocity, d_velocity, bytes));
    ok(nbody_memcpy_d2h(accelerations, d_accelerations, bytes));
    ok(nbody_free_device(d_bodies));
    ok(nbody_free_device(d_velocity));
    ok(nbody_free_device(d_accelerations));
    }
    if (!dump.empty()) {
        if (!write_file(dump + ".x.f4", bodies, bytes) || !write_file(dump + ".v.f4", velocity, bytes) ||
            !write_file(dump + ".a.f4", accelerations, bytes))
            die("cannot write state " + dump);
        char hdr[256];
        std::snprintf(hdr, sizeof hdr, "{\"n\": %d, \"steps_done\": %ld, \"dt\": %.9g, \"eps2\": %.9g, \"dtype\": \"%s\", \"ngpu\": %d, "
                      "\"layout\": \"float4 x,y,z,w LE (.f4)%s\"}\n",
                      n, steps_done + steps, dt, eps2, f64 ? "f64" : "f32", ngpu, f64 ? "; double4 x,y,z,w LE (.f8)" : "");
        if (!write_file(dump + ".json", hdr, std::strlen(hdr))) die("cannot write " + dump + ".json");
    }
    if (json && clock && clk.launches > 0)   // what clock the line below was measured at: cycles = the code, sclk = the box
        std::printf("{\"clock\": {\"force_kernel_ms_per_launch\": %.6f, \"launches\": %d, \"kernel_cycles_per_launch\": %.6g, \"sclk_mhz_under_load\": %.1f, "
                    "\"sclk_mhz_slowest_xcd\": %.1f, \"sclk_mhz_fastest_xcd\": %.1f, \"ms_per_launch_by_device_clock\": %.6f}}\n",
                    force_launches ? force_ms / force_launches : 0.0, clk.launches, clk.cycles_per_launch, clk.sclk_mhz, clk.sclk_mhz_min_xcd, clk.sclk_mhz_max_xcd,
                    clk.ticks_per_launch * 1e-5);
    if (json) {
        const double pairs = (double)n * (double)n * steps;
        std::printf("{\"n\": %d, \"steps\": %d, \"dt\": %.9g, \"eps2\": %.9g, \"kernel\": \"%s\", \"ngpu\": %d, \"seconds\": %.6f, \"pairs_per_s\": %.6g, "
                    "\"gflops_at_20\": %.6g, \"inplace_fallback_waves\": %llu, \"autotuned_choice\": %d, \"body0\": [%.9g, %.9g, %.9g, %.9g]}\n",
                    n, steps, dt, eps2, kname, ngpu, secs, secs > 0 ? pairs / secs : 0.0,
                    secs > 0 ? 20.0 * pairs / secs / 1e9 : 0.0, inplace_fallback_waves, autotuned_choice, n ? bodies[0].x : 0.f, n ? bodies[0].y : 0.f, n ? bodies[0].z : 0.f,
                    n ? bodies[0].w : 0.f);
    }
    ok(nbody_free_host(bodies));
    ok(nbody_free_host(velocity));
    ok(nbody_free_host(accelerations));
    return 0;
}
