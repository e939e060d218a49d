// ptrender — render a scene file of the reference's schema on the GPU and write the image the reference's
// window would show.  The reference (src/main.rs) takes the scene path as its only argument, renders
// scene.camera.samples samples per pixel and displays the tone-mapped buffer in a winit/wgpu window; this
// harness does the same through include/ptcore.hpp and writes files instead of opening a window:
//
//   ptrender scene.toml out [--seed S] [--spp N] [--device D] [--chunk K [--stop-after M] [--resume]]
//     out.png   what the Bgra8Unorm surface holds after renderer.rs:297-309 + shader.wgsl:32-44 (NaN pixels black)
//     out.pfm   the raw XYZ buffer (main.rs:102-107), little-endian PFM, bottom row first
//
// Checkpoint / resume (no counterpart in the reference, which keeps nothing): with --chunk K the render runs as sample ranges
// [0, K), [K, 2K) ... of PT_RENDER_SUM, the raw sums are added on the host and out.ckpt (header + W*H*4 floats) is rewritten
// after every range; --resume picks a render up from out.ckpt (same scene size, sample count, seed and chunk or it is
// refused) and --stop-after M ends the program once M samples are in the checkpoint (exit status 3: incomplete).  Sample s
// of pixel p draws from stream (seed, p, s) whatever ran before, so a resumed render is the uninterrupted chunked render bit
// for bit.
//
// Test/demo harness, not the product: the product is libptcore.so behind include/ptcore.h.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../include/ptcore.hpp"

namespace {

uint32_t crc32_of(const uint8_t* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }

void chunk(std::vector<uint8_t>& png, const char tag[5], const std::vector<uint8_t>& data) {
    put_be32(png, (uint32_t)data.size());
    const size_t at = png.size();
    png.insert(png.end(), tag, tag + 4);
    png.insert(png.end(), data.begin(), data.end());
    put_be32(png, crc32_of(&png[at], png.size() - at));
}

// RGBA8 PNG with stored (uncompressed) deflate blocks: no dependency, any viewer reads it
void write_png(const std::string& path, const std::vector<uint8_t>& bgra, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (1 + 4 * (size_t)w));
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t* p = &bgra[4 * ((size_t)y * w + x)];
            raw.push_back(p[2]); raw.push_back(p[1]); raw.push_back(p[0]); raw.push_back(255);
        }
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t len = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + len >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(len & 255)); z.push_back((uint8_t)(len >> 8));
        z.push_back((uint8_t)(~len & 255)); z.push_back((uint8_t)((~len >> 8) & 255));
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + len));
        if (raw.empty()) break;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'}, ihdr;
    put_be32(ihdr, w); put_be32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});
    chunk(png, "IHDR", ihdr);
    chunk(png, "IDAT", z);
    chunk(png, "IEND", {});
    std::ofstream(path, std::ios::binary).write((const char*)png.data(), (std::streamsize)png.size());
}

void write_pfm(const std::string& path, const std::vector<float>& xyzw, uint32_t w, uint32_t h) {
    std::ofstream f(path, std::ios::binary);
    f << "PF\n" << w << " " << h << "\n-1.0\n";
    for (uint32_t y = h; y-- > 0;)
        for (uint32_t x = 0; x < w; ++x) f.write((const char*)&xyzw[4 * ((size_t)y * w + x)], 12);
}

struct CheckpointHeader {          // out.ckpt: this header, then width * height * 4 floats of raw sums (w = samples so far)
    char magic[8];                  // "PTCKPT1\0"
    uint32_t width, height, samples_total, samples_done, chunk, sums_crc;
    uint64_t seed;
};

bool save_checkpoint(const std::string& path, const CheckpointHeader& h, const std::vector<float>& sums) {
    const std::string tmp = path + ".tmp";
    {
        std::ofstream f(tmp, std::ios::binary);
        f.write((const char*)&h, sizeof(h));
        f.write((const char*)sums.data(), (std::streamsize)(sums.size() * sizeof(float)));
        if (!f) return false;
    }
    return std::rename(tmp.c_str(), path.c_str()) == 0;       // a crash mid-write leaves the previous checkpoint intact
}

bool load_checkpoint(const std::string& path, CheckpointHeader& h, std::vector<float>& sums, std::string& why) {
    std::ifstream f(path, std::ios::binary);
    if (!f || !f.read((char*)&h, sizeof(h))) { why = "cannot read " + path; return false; }
    if (std::memcmp(h.magic, "PTCKPT1", 8) != 0) { why = path + " is not a ptrender checkpoint"; return false; }
    if ((uint64_t)h.width * h.height > (1ull << 28)) { why = "checkpoint header is damaged"; return false; }
    sums.resize((size_t)h.width * h.height * 4);
    if (!f.read((char*)sums.data(), (std::streamsize)(sums.size() * sizeof(float)))) { why = path + " is truncated"; return false; }
    if (crc32_of((const uint8_t*)sums.data(), sums.size() * sizeof(float)) != h.sums_crc) { why = path + " fails its checksum"; return false; }
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: ptrender scene.toml out_prefix [--seed S] [--spp N] [--device D] [--chunk K [--stop-after M] [--resume]]\n");
        return 2;
    }
    uint64_t seed = 1;
    long spp = -1, chunk_spp = 0, stop_after = -1;
    int device = 0;
    bool resume = false;
    for (int i = 3; i < argc; ++i) {
        const std::string k = argv[i];
        if (k == "--resume") { resume = true; continue; }
        if (i + 1 >= argc) { std::fprintf(stderr, "option %s needs a value\n", k.c_str()); return 2; }
        const char* v = argv[++i];
        if (k == "--seed") seed = std::strtoull(v, nullptr, 10);
        else if (k == "--spp") spp = std::strtol(v, nullptr, 10);
        else if (k == "--device") device = std::atoi(v);
        else if (k == "--chunk") chunk_spp = std::strtol(v, nullptr, 10);
        else if (k == "--stop-after") stop_after = std::strtol(v, nullptr, 10);
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    if ((resume || stop_after >= 0) && chunk_spp <= 0) { std::fprintf(stderr, "--resume and --stop-after need --chunk K\n"); return 2; }
    try {
        using namespace pathtracer;
        Device dev(device);
        const auto t0 = std::chrono::steady_clock::now();
        Scene scene = Scene::load(dev, argv[1]);
        const auto t1 = std::chrono::steady_clock::now();
        const uint32_t w = scene.camera.resolution[0], h = scene.camera.resolution[1];
        const uint32_t samples = spp > 0 ? (uint32_t)spp : scene.camera.samples;
        std::vector<float> xyzw;
        pt_stats st{};
        if (chunk_spp <= 0) {
            xyzw = scene.render(scene.camera, samples, seed);
            st = dev.stats();
        } else {
            const std::string ckpt = std::string(argv[2]) + ".ckpt";
            CheckpointHeader hd{};
            std::memcpy(hd.magic, "PTCKPT1", 8);
            hd.width = w; hd.height = h; hd.samples_total = samples; hd.samples_done = 0; hd.chunk = (uint32_t)chunk_spp; hd.seed = seed;
            std::vector<float> sums((size_t)w * h * 4, 0.0f);
            if (resume) {
                CheckpointHeader old{};
                std::string why;
                if (!load_checkpoint(ckpt, old, sums, why)) { std::fprintf(stderr, "ptrender: --resume: %s\n", why.c_str()); return 1; }
                if (old.width != w || old.height != h || old.samples_total != samples || old.seed != seed || old.chunk != hd.chunk ||
                    old.samples_done > samples || old.samples_done % hd.chunk != 0u) {
                    std::fprintf(stderr, "ptrender: --resume: %s belongs to another render (%ux%u, %u spp, seed %llu, chunk %u)\n", ckpt.c_str(),
                                 old.width, old.height, old.samples_total, (unsigned long long)old.seed, old.chunk);
                    return 1;
                }
                hd.samples_done = old.samples_done;
                std::printf("resuming at sample %u of %u\n", hd.samples_done, samples);
            }
            while (hd.samples_done < samples) {
                if (stop_after >= 0 && hd.samples_done >= (uint32_t)stop_after) {
                    std::printf("stopped with %u of %u samples in %s\n", hd.samples_done, samples, ckpt.c_str());
                    return 3;
                }
                const uint32_t count = std::min<uint32_t>(hd.chunk, samples - hd.samples_done);
                const std::vector<float> part = scene.render_range(scene.camera, hd.samples_done, count, samples, seed, PT_RENDER_SUM);
                const pt_stats cs = dev.stats();
                st.total_ms += cs.total_ms; st.trace_ms += cs.trace_ms; st.trace_rays += cs.trace_rays; st.paths += cs.paths; st.segments += cs.segments;
                for (size_t i = 0; i < sums.size(); ++i) sums[i] += part[i];
                hd.samples_done += count;
                hd.sums_crc = crc32_of((const uint8_t*)sums.data(), sums.size() * sizeof(float));
                if (!save_checkpoint(ckpt, hd, sums)) { std::fprintf(stderr, "ptrender: cannot write %s\n", ckpt.c_str()); return 1; }
            }
            xyzw = sums;
            finalize_sums(xyzw);
        }
        const auto t2 = std::chrono::steady_clock::now();
        const Display d = tonemap(dev, xyzw, w, h, 5.0f, 10.0f, PT_TONEMAP_SKIP_NONFINITE);
        write_png(std::string(argv[2]) + ".png", d.bgra8, w, h);
        write_pfm(std::string(argv[2]) + ".pfm", xyzw, w, h);
        const auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::printf("Loaded %u triangles in %.1f ms; rendered %ux%u, %u spp in %.3f seconds (device %.1f ms, %.2f segments per path, "
                    "traversal %.0f Mrays/s); average luminance %.4g -> %s.png, %s.pfm\n",
                    scene.info().n_triangles, ms(t0, t1), w, h, samples, ms(t1, t2) / 1e3, st.total_ms,
                    (double)st.segments / (double)std::max<uint64_t>(st.paths, 1), st.trace_rays / std::max(st.trace_ms, 1e-9) / 1e3,
                    d.avg_luminance, argv[2], argv[2]);
    } catch (const pathtracer::Error& e) {
        std::fprintf(stderr, "ptrender: error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
