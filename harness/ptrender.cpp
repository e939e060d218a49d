// ptrender — render a scene file of the reference's schema on the GPU and write the image the reference's
// window would show.  The reference (src/main.rs) takes the scene path as its only argument, renders
// scene.camera.samples samples per pixel and displays the tone-mapped buffer in a winit/wgpu window; this
// harness does the same through include/ptcore.hpp and writes files instead of opening a window:
//
//   ptrender scene.toml out [--seed S] [--spp N] [--device D] [--lbvh]
//     out.png   what the Bgra8Unorm surface holds after renderer.rs:297-309 + shader.wgsl:32-44 (NaN pixels black)
//     out.pfm   the raw XYZ buffer (main.rs:102-107), little-endian PFM, bottom row first
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

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: ptrender scene.toml out_prefix [--seed S] [--spp N] [--device D]\n");
        return 2;
    }
    uint64_t seed = 1;
    long spp = -1;
    int device = 0;
    for (int i = 3; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--seed") seed = std::strtoull(argv[i + 1], nullptr, 10);
        else if (k == "--spp") spp = std::strtol(argv[i + 1], nullptr, 10);
        else if (k == "--device") device = std::atoi(argv[i + 1]);
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    try {
        using namespace pathtracer;
        Device dev(device);
        const auto t0 = std::chrono::steady_clock::now();
        Scene scene = Scene::load(dev, argv[1]);
        const auto t1 = std::chrono::steady_clock::now();
        const uint32_t w = scene.camera.resolution[0], h = scene.camera.resolution[1];
        const uint32_t samples = spp > 0 ? (uint32_t)spp : scene.camera.samples;
        const std::vector<float> xyzw = scene.render(scene.camera, samples, seed);
        const auto t2 = std::chrono::steady_clock::now();
        const pt_stats st = dev.stats();
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
