// hanamaru-hip — host driver with the reference binary's flag surface and outputs (main.rs:1226-1295,
// renderer.rs:205-251): `hanamaru-hip -w W -h H -s S -t SEC -i SEC`.  Stand-in for the Rust host (no Rust
// toolchain here): scene authoring + PNG writing stay on the host, the render loop calls the C ABI.
// Additive flags (do not change defaults): --scene NAME, --assets DIR, --batch N, --launch L, --inflight K, --precise, --gpus N / --gpu-ids LIST,
// --checkpoint FILE (write the fp32 accumulator + sampling count when the render stops) and --resume FILE
// (continue from such a file: samplings are independent and seeded by index, so a resumed render adds exactly
// the samplings that are missing — SURVEY.md §8f rank 3; the reference has no resumable state).
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hanamaru_hip.h"
#include "hanamaru_host.h"

static FILE *g_log = nullptr;
static void tee(const char *fmt, ...) {  // main.rs:47-51
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    printf("%s\n", buf);
    if (g_log) { fputs(buf, g_log); fputc('\n', g_log); }
}
static double now_sec() {
    using namespace std::chrono;
    return duration_cast<duration<double>>(steady_clock::now().time_since_epoch()).count();
}
#define CHECK_HR(expr)                                                          \
    do {                                                                        \
        int rc_ = (expr);                                                       \
        if (rc_ != 0) { fprintf(stderr, "%s: %s\n", #expr, hr_last_error()); return 1; } \
    } while (0)

static void usage(const char *prog) {
    printf("Usage: %s [options]\n\nOptions:\n"
           "        --help          print this help menu\n"
           "    -d, --debug         use debug mode\n"
           "    -w, --width WIDTH   output resolution width\n"
           "    -h, --height HEIGHT output resolution height\n"
           "    -s, --sampling SAMPLING\n                        sampling limit\n"
           "    -t, --time TIME     time limit sec\n"
           "    -i, --interval INTERVAL\n                        report interval sec\n"
           "        --scene NAME    rtcamp6_v3_1 (default) | rtcamp6_v3 | rtcamp6_v2 | rtcamp6_v1 | rtcamp5 | tbf3 | material_examples | simple |\n"
           "                        spheres | rtcamp6_dodeca | cornell_mini\n"
           "        --assets DIR    directory holding models/ and textures/ (default: ./assets, then .)\n"
           "        --batch N       samplings per progress report (default 1: one \"rendering:\" line per sampling, as the reference)\n"
           "        --launch L      reports per GPU launch (default 0: as many as fill the chip — 4 samplings per device at 1920x1080); the lines of a\n"
           "                        launch are printed when it is done, its time split evenly over them; 1 = a launch per report\n"
           "        --inflight K    launches enqueued ahead on the GPU (default 8; 1 = wait for every launch before the next starts)\n"
           "        --precise       precise shading: the geometry of every bounce in f64 from the reference's f64 draws (hr_set_option\n"
           "                        \"precise_shading\"): the reference's own arithmetic on refraction chains, small spheres and roughness maps;\n"
           "                        default: on for scenes without meshes (where it costs 2 - 4 %%), off for mesh scenes (7 - 30 %%);\n"
           "                        --no-precise: fp32 shading whatever the scene\n"
           "        --gpus N        render on devices 0..N-1 of this node from this one process: device r takes every N-th sampling,\n"
           "                        the accumulators are summed with one RCCL all-reduce when an image is written (default 1)\n"
           "        --gpu-ids LIST  the same with an explicit comma-separated device list\n"
           "        --checkpoint F  write accumulator + sampling count to F when the render stops\n"
           "        --resume F      continue from a checkpoint file\n",
           prog);
}

int main(int argc, char **argv) {
    uint32_t width = 1920, height = 1080, sampling = 1000;  // main.rs:1249-1251
    double time_limit = 123.0, interval = 15.0;              // main.rs:1255-1256
    std::string scene_name = "rtcamp6_v3_1", assets, ckpt_out, ckpt_in, gpu_ids;
    int gpus = 1;
    int batch = 1;      // samplings per report_progress call ("rendering:" line); 1 = the reference's cadence
    int launch = 0;     // reports per GPU launch; 0 = as many as fill the chip (4 samplings per device at 1920x1080)
    int inflight = 8;   // launches enqueued ahead of the one being reported
    bool debug = false;
    int precise = -1;    // option precise_shading: -1 = the library's automatic choice
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "Argument to option '%s' missing.\n", name); exit(1); }
            return argv[++i];
        };
        if (a == "--help") { usage(argv[0]); return 0; }
        else if (a == "-d" || a == "--debug") debug = true;
        else if (a == "-w" || a == "--width") width = (uint32_t)strtoul(val("w"), nullptr, 10);
        else if (a == "-h" || a == "--height") height = (uint32_t)strtoul(val("h"), nullptr, 10);
        else if (a == "-s" || a == "--sampling") sampling = (uint32_t)strtoul(val("s"), nullptr, 10);
        else if (a == "-t" || a == "--time") time_limit = strtod(val("t"), nullptr);
        else if (a == "-i" || a == "--interval") interval = strtod(val("i"), nullptr);
        else if (a == "--scene") scene_name = val("scene");
        else if (a == "--assets") assets = val("assets");
        else if (a == "--batch") batch = atoi(val("batch"));
        else if (a == "--launch") launch = atoi(val("launch"));
        else if (a == "--inflight") inflight = atoi(val("inflight"));
        else if (a == "--precise") precise = 1;
        else if (a == "--no-precise") precise = 0;
        else if (a == "--gpus") gpus = atoi(val("gpus"));
        else if (a == "--gpu-ids") gpu_ids = val("gpu-ids");
        else if (a == "--checkpoint") ckpt_out = val("checkpoint");
        else if (a == "--resume") ckpt_in = val("resume");
        else { fprintf(stderr, "Unrecognized option: '%s'.\n", a.c_str()); return 1; }
    }
    if (batch < 1) { fprintf(stderr, "--batch must be at least 1.\n"); return 1; }
    if (launch < 0) { fprintf(stderr, "--launch must be at least 1 (or 0: automatic).\n"); return 1; }
    if (inflight < 1) { fprintf(stderr, "--inflight must be at least 1.\n"); return 1; }
    if (width == 0 || height == 0) { fprintf(stderr, "width and height must be positive.\n"); return 1; }
    if (gpus < 1) { fprintf(stderr, "--gpus must be at least 1.\n"); return 1; }
    if (assets.empty()) {
        FILE *probe = fopen("assets/models/box.obj", "rb");
        if (probe) { fclose(probe); assets = "assets"; } else assets = ".";
    }
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);   // dmabuf IPC (RCCL on this driver), unless the caller says otherwise; before the HIP runtime starts
    g_log = fopen("result.txt", "w");
    double total_begin = now_sec();
    tee("num threads: %d.", 1);  // main.rs:1261 prints rayon's pool size; here: one host thread drives one GPU
    tee("resolution: %ux%u.", width, height);
    tee("max sampling: %ux%u spp.", sampling, 4u);
    tee("time limit: %.2f sec.", time_limit);
    tee("report interval: %.2f sec.", interval);

    double init_begin = now_sec();
    hh_scene *scene = nullptr;
    if (hh_scene_create(scene_name.c_str(), assets.c_str(), &scene) != 0) { fprintf(stderr, "scene: %s\n", hh_last_error()); return 1; }
    // devices: samplings are independent and seeded by index, so device r of N renders samplings first + r, first + r + N, ...
    // (bench.py does the same with one process per GPU and an RCCL all-reduce)
    std::vector<int> devices;
    if (!gpu_ids.empty()) {
        for (size_t p = 0; p < gpu_ids.size();) {
            size_t q = gpu_ids.find(',', p);
            devices.push_back(atoi(gpu_ids.substr(p, q == std::string::npos ? std::string::npos : q - p).c_str()));
            if (q == std::string::npos) break;
            p = q + 1;
        }
    } else {
        for (int d = 0; d < (gpus > 0 ? gpus : 1); d++) devices.push_back(d);
    }
    const uint32_t ndev = (uint32_t)devices.size();
    std::vector<hr_ctx *> ctxs(ndev, nullptr);
    for (uint32_t r = 0; r < ndev; r++) {
        CHECK_HR(hr_create(devices[r], &ctxs[r]));
        CHECK_HR(hr_upload_scene(ctxs[r], hh_scene_desc(scene)));
        CHECK_HR(hr_set_resolution(ctxs[r], width, height));
        if (precise >= 0) CHECK_HR(hr_set_option(ctxs[r], "precise_shading", (double)precise));
    }
    hr_ctx *ctx = ctxs[0];
    if (ndev > 1) tee("devices: %u.", ndev);
    tee("init scene: %.2f sec.", now_sec() - init_begin);
    // The image needs the sum of all devices' accumulators: ONE all-reduce over RCCL (hr_allreduce_accumulators — the sum lands
    // in a separate buffer per device, so every device keeps accumulating its own samplings afterwards); hr_resolve /
    // hr_read_accumulator on device 0 then see the total (renderer.rs:64-90 runs after the sum).
    // A multi-GPU box whose RCCL cannot be loaded (hr_comm_init_local: HR_ERR_UNSUPPORTED) still renders: the host then sums the
    // devices' accumulators itself when an image or a checkpoint is written (`sum_acc` below) — slower, same result.
    bool host_sum = false;
    if (ndev > 1) {
        int rc = hr_comm_init_local(ctxs.data(), (int)ndev);
        if (rc == HR_ERR_UNSUPPORTED) { host_sum = true; tee("no RCCL (%s): accumulators are summed on the host.", hr_last_error()); }
        else if (rc != 0) { fprintf(stderr, "hr_comm_init_local: %s\n", hr_last_error()); return 1; }
    }
    std::vector<float> sum_acc, part;
    auto combine = [&]() -> int {
        if (ndev == 1) return 0;
        if (!host_sum) return hr_allreduce_accumulators(ctxs.data(), (int)ndev) != 0;
        sum_acc.resize((size_t)width * height * 3);
        part.resize(sum_acc.size());
        if (hr_read_accumulator(ctxs[0], sum_acc.data()) != 0) return 1;
        for (uint32_t r = 1; r < ndev; r++) {
            if (hr_read_accumulator(ctxs[r], part.data()) != 0) return 1;
            for (size_t i = 0; i < sum_acc.size(); i++) sum_acc[i] += part[i];
        }
        return 0;
    };
    // host-side sum: device 0 resolves the total from its accumulator, then gets its own partial sums back
    auto resolve = [&](uint32_t s, uint8_t *out) -> int {
        if (!host_sum) return hr_resolve(ctx, s, out);
        if (hr_read_accumulator(ctx, part.data()) != 0 || hr_write_accumulator(ctx, sum_acc.data()) != 0) return 1;
        int rc = hr_resolve(ctx, s, out);
        return hr_write_accumulator(ctx, part.data()) != 0 ? 1 : rc;
    };

    std::vector<uint8_t> rgb((size_t)width * height * 3);
    double begin = now_sec(), last_progress = begin, last_image = begin;
    uint32_t counter = 0, sampled = 0;
    std::string last_png;     // the image file save() wrote last: result.png is a copy of the final one (one PNG encode, not two)
    auto save = [&](uint32_t s) -> int {
        char path[32];
        snprintf(path, sizeof path, "%03u.png", counter);
        double t0 = now_sec();
        if (combine() || resolve(s, rgb.data()) != 0) { fprintf(stderr, "hr_resolve: %s\n", hr_last_error()); return 1; }
        printf("update_imgbuf: %.3f sec\n", now_sec() - t0);
        last_png = path;
        return hh_write_png_rgb8(path, rgb.data(), width, height);
    };
    uint32_t first = 1;
    // checkpoint = {magic, width, height, samplings done, FNV-1a of the scene name} + the fp32 accumulator
    const uint32_t CKPT_MAGIC = 0x32415248u;   // "HRA2"
    uint32_t scene_hash = 2166136261u;
    for (char ch : scene_name) scene_hash = (scene_hash ^ (uint8_t)ch) * 16777619u;
    if (!ckpt_in.empty()) {
        FILE *f = fopen(ckpt_in.c_str(), "rb");
        uint32_t hdr[5] = {0, 0, 0, 0, 0};
        std::vector<float> acc((size_t)width * height * 3);
        bool ok = f && fread(hdr, 4, 5, f) == 5 && hdr[0] == CKPT_MAGIC && hdr[1] == width && hdr[2] == height && hdr[4] == scene_hash &&
                  fread(acc.data(), sizeof(float), acc.size(), f) == acc.size();
        if (f) fclose(f);
        if (!ok) { fprintf(stderr, "cannot resume from %s (missing, wrong magic, resolution or scene)\n", ckpt_in.c_str()); return 1; }
        CHECK_HR(hr_write_accumulator(ctx, acc.data()));
        first = hdr[3] + 1;
        sampled = hdr[3];
        printf("resumed at %ux4 sampled\n", hdr[3]);
    }
    if (debug) {  // main.rs:1279-1281: DebugRenderer { mode: FocalPlane }, max_sampling 1, report_progress = update_imgbuf + stop
        CHECK_HR(hr_render_debug(ctx, 3));
        CHECK_HR(hr_synchronize(ctx));
        CHECK_HR(hr_resolve(ctx, 1, rgb.data()));
        sampled = 1;
        sampling = 0;
    }
    // Renderer::render's loop with report_progress (renderer.rs:32-43, 205-251).  What is REPORTED and what is LAUNCHED are two things:
    //   * a REPORT = `--batch` samplings = one "rendering:" line; the default, --batch 1, is the reference's own cadence: a line per sampling;
    //   * the GPU is fed LAUNCHES of whole reports — `--launch` of them, by default as many as fill the chip (4 samplings per device at
    //     1920x1080: one kernel launch per sampling costs 3 - 8 % of the rate) — and up to --inflight launches are enqueued ahead
    //     (hr_render only enqueues; hr_mark behind every launch, hr_wait for the oldest), so the GPU never drains between progress lines.
    //     When a launch is done its reports are printed together: the launch's wall time is split EVENLY over their lines (said once in the log).
    // report_progress's three rules keep their order and their meaning:
    //   * time limit (renderer.rs:222-231: stop when used + 1.1 x last > limit, `last` = seconds per report).  The reference asks this after
    //     a sampling, before it starts the next; here the question is asked when reports are ISSUED, for the moment they would finish: n
    //     reports are issued only if used + 1.1 x last x (reports in flight + n) <= limit, and a launch shrinks to the n that still fits
    //     (down to one report).  With one report in flight at a time (-i 0, or --launch 1 --inflight 1) that is the reference's rule to the
    //     letter.  When everything in flight has been reported and nothing may follow, the final image is written: "reached time limit"
    //     if the rule says so, else "reached max sampling" (in that order, renderer.rs:222-241).
    //   * progress image (renderer.rs:243-251): asked at launch boundaries (the accumulator holds whole launches); when the interval has
    //     passed, the launches in flight are awaited and reported first, so that the image holds exactly the samplings of the
    //     "rendering:" line before it, as in the reference (one pipeline drain per image).  An interval of 0 asks for an image after
    //     every report: launches are then one report long, one in flight, and `-s 5 -i 0` prints and writes exactly what the
    //     reference does — 000.png .. 003.png after samplings 1 .. 4, final 004.png.
    struct Launch { uint32_t begin, end; double issued; std::vector<uint64_t> ticket; };
    std::vector<Launch> q;     // issued, not yet reported; oldest first
    uint32_t next_s = first;
    const uint32_t B = (uint32_t)batch;
    uint32_t lrep = launch > 0 ? (uint32_t)launch : 0;   // reports per launch
    if (!lrep) {   // as many reports as make the library's own automatic launch size (hanamaru_hip.h "batch": about 33 M paths) on every device
        const uint64_t per_sampling = ((uint64_t)(width + 3) / 4) * ((height + 3) / 4) * 64u;
        const uint64_t lsize = std::min<uint64_t>(64, std::max<uint64_t>(4, (33177600ull + per_sampling - 1) / per_sampling));
        lrep = (uint32_t)std::max<uint64_t>(1, lsize * ndev / B);
    }
    if (interval <= 0.0) lrep = 1;
    if (lrep > 1 && !debug) printf("launches of %u reports (%u samplings): a launch's time is split evenly over its reports' lines.\n", lrep, lrep * B);
    uint32_t in_flight = 0;   // reports issued, not yet printed
    auto reports_of = [&](uint32_t b, uint32_t e) { return (e - b + B - 1) / B; };
    auto issue = [&](uint32_t nrep) -> int {
        Launch c;
        c.begin = next_s;
        c.end = (uint64_t)next_s + (uint64_t)nrep * B > (uint64_t)sampling + 1 ? sampling + 1 : next_s + nrep * B;
        c.issued = now_sec();
        c.ticket.assign(ndev, 0);
        for (uint32_t r = 0; r < ndev; r++) {
            // device r of N renders the samplings with (s - 1) mod N == r, whatever the launch's first sampling is — a launch
            // shorter than N leaves some devices without work, their marker is then reached at once
            const uint32_t b = c.begin + (r + ndev - (c.begin - 1u) % ndev) % ndev;
            if (hr_render(ctxs[r], b, c.end, ndev) != 0 || hr_mark(ctxs[r], &c.ticket[r]) != 0) { fprintf(stderr, "hr_render: %s\n", hr_last_error()); return 1; }
        }
        next_s = c.end;
        in_flight += reports_of(c.begin, c.end);
        q.push_back(c);
        return 0;
    };
    bool measured = false;    // has any launch been reported?  (a measured zero is a measurement)
    double last = 0.0;        // `from_last_sampling_sec`: seconds per report of the last launch reported
    double used = 0.0;
    // wait for the oldest launch in flight and print its reports' lines (renderer.rs:206-214)
    auto report = [&]() -> int {
        const Launch c = q.front();
        q.erase(q.begin());
        for (uint32_t r = 0; r < ndev; r++)
            if (hr_wait(ctxs[r], c.ticket[r]) != 0) { fprintf(stderr, "hr_wait: %s\n", hr_last_error()); return 1; }
        const double now = now_sec();
        const uint32_t n = reports_of(c.begin, c.end);
        // the launch's own time: from the previous report (the pipeline is full: launches finish back to back), or from its issue if that is later
        const double t0 = std::max(last_progress, c.issued);
        last = (now - t0) / (double)n;
        measured = true;
        for (uint32_t j = 0; j < n; j++) {
            sampled = std::min(c.begin + (j + 1) * B, c.end) - 1;
            used = t0 + last * (double)(j + 1) - begin;
            printf("rendering: %ux4 sampled (last %.3f sec). total: %.3f sec (%.2f %%).\n", sampled, last, used, used / time_limit * 100.0);
        }
        in_flight -= n;
        last_progress = now;
        used = now - begin;
        return 0;
    };
    // renderer.rs:222-241: the final image takes the current counter
    auto finish = [&](const char *why) -> int {
        for (uint32_t r = 0; r < ndev; r++)
            if (hr_synchronize(ctxs[r]) != 0) { fprintf(stderr, "hr_synchronize: %s\n", hr_last_error()); return 1; }
        printf("%s\n", why);
        printf("output final image: %03u.png\n", counter);
        printf("remain: %.3f sec.\n", time_limit - used);
        return save(sampled);
    };
    // how many reports may be enqueued now?  (samplings left, room in the pipeline, and the time-limit rule asked for the moment they would finish)
    const size_t depth = interval <= 0.0 ? 1 : (size_t)inflight;
    auto may_issue = [&]() -> uint32_t {
        if (next_s > sampling || q.size() >= depth) return 0;
        const uint32_t n = std::min<uint32_t>(lrep, reports_of(next_s, sampling + 1));
        if (!measured) return n;   // nothing measured yet: fill the pipeline
        const double room = time_limit - (now_sec() - begin);
        const double fit = last > 0.0 ? room / (1.1 * last) - (double)in_flight : (room >= 0.0 ? (double)n : 0.0);
        if (fit < 1.0) return 0;
        return fit < (double)n ? (uint32_t)fit : n;
    };
    if (first > sampling && !debug && sampled > 0) {   // resumed from a checkpoint that already holds every requested sampling: just resolve it
        printf("reached max sampling\n");
        if (save(sampled)) return 1;
    }
    bool running = first <= sampling;
    while (running) {
        for (uint32_t n; (n = may_issue()) != 0;) if (issue(n)) return 1;
        if (q.empty()) {   // nothing in flight and nothing may follow: the render ends here (renderer.rs:222-241, the time limit asked first)
            // (samplings left over: only the time-limit rule can have refused them)
            if (finish(next_s <= sampling || used + 1.1 * last > time_limit ? "reached time limit" : "reached max sampling")) return 1;
            break;
        }
        if (report()) return 1;
        if (last_progress - last_image >= interval) {   // renderer.rs:243-251, with the `now` of the report
            while (!q.empty()) if (report()) return 1;   // the launches in flight: the image then holds exactly the samplings reported
            // nothing is in flight now: the reference's own rules apply as they stand, in their order (renderer.rs:222-241)
            if (used + 1.1 * last > time_limit) { if (finish("reached time limit")) return 1; break; }
            if (sampled >= sampling) { if (finish("reached max sampling")) return 1; break; }
            for (uint32_t r = 0; r < ndev; r++) CHECK_HR(hr_synchronize(ctxs[r]));
            printf("output progress image: %03u.png\n", counter);
            if (save(sampled)) return 1;
            counter++;
            last_image = last_progress;             // `now` of the report that triggered it (renderer.rs:250)
        }
    }
    for (uint32_t r = 0; r < ndev; r++) CHECK_HR(hr_synchronize(ctxs[r]));
    if (!ckpt_out.empty()) {
        std::vector<float> acc((size_t)width * height * 3);
        if (combine()) { fprintf(stderr, "checkpoint: %s\n", hr_last_error()); return 1; }
        if (host_sum) acc = sum_acc;
        else CHECK_HR(hr_read_accumulator(ctx, acc.data()));
        uint32_t hdr[5] = {CKPT_MAGIC, width, height, sampled, scene_hash};
        FILE *f = fopen(ckpt_out.c_str(), "wb");
        bool ok = f && fwrite(hdr, 4, 5, f) == 5 && fwrite(acc.data(), sizeof(float), acc.size(), f) == acc.size();
        if (f) fclose(f);
        if (!ok) { fprintf(stderr, "cannot write checkpoint %s\n", ckpt_out.c_str()); return 1; }
    }
    {   // main.rs:1217: result.png = the final image — the bytes finish() has just written as NNN.png
        bool copied = false;
        if (!last_png.empty()) {
            FILE *in = fopen(last_png.c_str(), "rb"), *out = in ? fopen("result.png", "wb") : nullptr;
            if (in && out) {
                std::vector<char> buf(1 << 20);
                size_t n;
                copied = true;
                while ((n = fread(buf.data(), 1, buf.size(), in)) > 0) copied = copied && fwrite(buf.data(), 1, n, out) == n;
                copied = copied && !ferror(in);
            }
            if (in) fclose(in);
            if (out) copied = (fclose(out) == 0) && copied;
        }
        if (!copied && hh_write_png_rgb8("result.png", rgb.data(), width, height) != 0) { fprintf(stderr, "png: %s\n", hh_last_error()); return 1; }
    }
    tee("sampled: %ux%u spp.", sampled, 4u);
    hr_stats st;
    if (hr_get_stats(ctx, &st) == 0) {
        double sec = st.trace_kernel_ms * 1e-3;
        uint64_t paths = st.paths;
        for (uint32_t r = 1; r < ndev; r++) { hr_stats o; if (hr_get_stats(ctxs[r], &o) == 0) paths += o.paths; }
        // two clocks: until the last sampling was reported (the render itself), and until here (+ the final hr_resolve and the PNG encoder)
        tee("gpu: %.3f Mpaths/s wall (%.3f incl. the final image), trace kernel %.3f s, seed kernel %.3f s.", (double)paths / std::max(1e-9, last_progress - begin) * 1e-6,
            (double)paths / (now_sec() - begin) * 1e-6, sec, st.seed_kernel_ms * 1e-3);
    }
    double total = now_sec() - total_begin;
    double used_percent = total / time_limit * 100.0;
    tee("total %g sec. used %.2f %% (x %.2f)", total, used_percent, 100.0 / used_percent);
    for (hr_ctx *c : ctxs) hr_destroy(c);
    hh_scene_destroy(scene);
    if (g_log) fclose(g_log);
    return 0;
}
