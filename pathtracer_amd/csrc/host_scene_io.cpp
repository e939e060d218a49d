// host_scene_io.cpp — scene ingestion, the step BEFORE the hot path: the
// reference's TOML scene file + OBJ/MTL meshes + PNG textures -> the arrays pt_scene_create takes.
// Host-side C++ mirror of
//
//   scene.rs:64-183    SceneFile / Camera / CameraTransform / Sky / Mesh (serde schema and defaults)
//   scene.rs:194-282   Scene::load: mesh.transform * camera_transform, object -> group -> poly order,
//                      quads split (a,b,c),(a,c,d), >4-gons rejected, material 0 = Material::default
//   scene.rs:16-62     ObjExt::triangle: uv raw, normals through transform_vector, positions through
//                      transform_point
//   material.rs:39-78,206-253   Material::from_mtl and the image -> texel conversions
//   color.rs:6-12,119-133       sRGB decode of texture pixels (flat Kd/Ke are taken as they are)
//
// nalgebra 0.31.3 semantics in binary32 with the same operation order: Matrix4::from
// ([[f32;4];4]) is column-major, look_at_rh, M*N by column axpy, transform_point divides by the
// homogeneous coordinate, transform_vector divides the INPUT by M[3,0..3].v when that is non-zero.
//
// Not reproduced: the `obj` crate accepts more of the OBJ grammar than this parser (v/vt/vn/f/o/g/
// usemtl/mtllib are read; triangles come out in file order, which is the crate's object -> group ->
// poly order whenever no object or group is re-opened); textures must be 8-bit non-interlaced PNG;
// of TOML, multi-line strings and date-times are not read (the schema has neither).
// Parity unpinned: the reference holds no vectors for its loader (tests/test_scene_io.py has
// hand-derived cases and checks this file against the Python restatement of the same rules).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ptcore.h"

int ptc_fail(struct pt_ctx* ctx, int code, const std::string& msg);

namespace {

[[noreturn]] void fail(const std::string& m) { throw std::runtime_error(m); }

std::string read_file(const std::string& path, bool binary) {
    std::ifstream f(path, binary ? std::ios::binary : std::ios::in);
    if (!f) fail(path + ": cannot open");
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

std::string dir_of(const std::string& path) {
    const size_t k = path.find_last_of('/');
    return k == std::string::npos ? std::string(".") : (k == 0 ? std::string("/") : path.substr(0, k));
}
std::string join_path(const std::string& dir, const std::string& name) {
    if (!name.empty() && name[0] == '/') return name;
    return dir + "/" + name;
}

// ---- TOML (the subset the schema can contain) -----------------------------------------------------

struct Value {
    enum Kind { Int, Float, Bool, String, Array, Table } kind = Table;
    int64_t i = 0;
    double f = 0.0;
    bool b = false;
    std::string s;
    std::vector<Value> a;
    std::vector<std::pair<std::string, Value>> t;       // insertion order
    bool array_of_tables = false;                       // an Array created by [[header]]

    const Value* get(const std::string& k) const {
        for (auto& kv : t) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    Value& child(const std::string& k) {
        for (auto& kv : t) if (kv.first == k) return kv.second;
        t.emplace_back(k, Value());
        return t.back().second;
    }
    bool is_number() const { return kind == Int || kind == Float; }
    double number() const { return kind == Int ? (double)i : f; }
};

class Toml {
public:
    explicit Toml(const std::string& text) : s_(text) {}
    Value parse() {
        Value root;
        Value* cur = &root;
        for (;;) {
            skip_ws_nl();
            if (eof()) break;
            if (peek() == '[') {
                ++p_;
                const bool aot = peek() == '[';
                if (aot) ++p_;
                std::vector<std::string> path = key_path();
                expect(']');
                if (aot) expect(']');
                end_of_line();
                cur = &root;
                for (size_t k = 0; k < path.size(); ++k) {
                    Value& c = cur->child(path[k]);
                    const bool last = k + 1 == path.size();
                    if (last && aot) {
                        if (c.kind == Value::Table && c.t.empty() && !c.array_of_tables) { c.kind = Value::Array; c.array_of_tables = true; }
                        if (c.kind != Value::Array || !c.array_of_tables) err("[[" + path[k] + "]] redefines a key");
                        c.a.emplace_back();
                        cur = &c.a.back();
                    } else if (c.kind == Value::Array && c.array_of_tables && !c.a.empty()) {
                        cur = &c.a.back();
                    } else if (c.kind == Value::Table) {
                        cur = &c;
                    } else err("[" + path[k] + "] redefines a key");
                }
            } else {
                std::vector<std::string> path = key_path();
                skip_ws();
                expect('=');
                skip_ws();
                Value v = value();
                end_of_line();
                assign(*cur, path, std::move(v));
            }
        }
        return root;
    }

private:
    const std::string& s_;
    size_t p_ = 0;
    int line_ = 1;

    bool eof() const { return p_ >= s_.size(); }
    char peek() const { return eof() ? '\0' : s_[p_]; }
    [[noreturn]] void err(const std::string& m) const { fail("TOML line " + std::to_string(line_) + ": " + m); }
    void expect(char c) { skip_ws(); if (peek() != c) err(std::string("expected '") + c + "'"); ++p_; }
    void skip_ws() { while (peek() == ' ' || peek() == '\t') ++p_; }
    void skip_comment() { if (peek() == '#') while (!eof() && peek() != '\n') ++p_; }
    void skip_ws_nl() {
        for (;;) {
            skip_ws();
            skip_comment();
            if (peek() == '\r') { ++p_; continue; }
            if (peek() == '\n') { ++p_; ++line_; continue; }
            break;
        }
    }
    void end_of_line() {
        skip_ws();
        skip_comment();
        if (peek() == '\r') ++p_;
        if (eof()) return;
        if (peek() != '\n') err("unexpected text after a value");
        ++p_; ++line_;
    }
    static bool bare(char c) { return isalnum((unsigned char)c) || c == '_' || c == '-'; }
    std::vector<std::string> key_path() {
        std::vector<std::string> path;
        for (;;) {
            skip_ws();
            if (peek() == '"' || peek() == '\'') path.push_back(string());
            else {
                size_t b = p_;
                while (bare(peek())) ++p_;
                if (p_ == b) err("expected a key");
                path.push_back(s_.substr(b, p_ - b));
            }
            skip_ws();
            if (peek() != '.') break;
            ++p_;
        }
        return path;
    }
    void assign(Value& tbl, const std::vector<std::string>& path, Value v) {
        Value* cur = &tbl;
        for (size_t k = 0; k + 1 < path.size(); ++k) {
            cur = &cur->child(path[k]);
            if (cur->kind != Value::Table) err("dotted key through a non-table");
        }
        if (cur->get(path.back())) err("duplicate key " + path.back());
        cur->t.emplace_back(path.back(), std::move(v));
    }
    std::string string() {
        const char q = peek();
        ++p_;
        if (peek() == q && p_ + 1 < s_.size() && s_[p_ + 1] == q) err("multi-line strings are not supported");
        std::string out;
        for (;;) {
            if (eof() || peek() == '\n') err("unterminated string");
            char c = s_[p_++];
            if (c == q) break;
            if (q == '"' && c == '\\') {
                char e = eof() ? '\0' : s_[p_++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case '"': out += '"'; break;
                    case '\\': out += '\\'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': case 'U': {
                        const int nd = e == 'u' ? 4 : 8;
                        if (p_ + nd > s_.size()) err("bad unicode escape");
                        unsigned long cp = strtoul(s_.substr(p_, nd).c_str(), nullptr, 16);
                        p_ += nd;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: err("bad escape in string");
                }
            } else out += c;
        }
        return out;
    }
    int depth_ = 0;
    struct Nest { int& d; explicit Nest(int& x) : d(x) { ++d; } ~Nest() { --d; } };
    Value value() {
        Nest nest(depth_);
        if (depth_ > 64) err("arrays / inline tables nested deeper than 64 levels");
        Value v;
        const char c = peek();
        if (c == '"' || c == '\'') { v.kind = Value::String; v.s = string(); return v; }
        if (c == '[') {
            ++p_;
            v.kind = Value::Array;
            for (;;) {
                skip_ws_nl();
                if (peek() == ']') { ++p_; break; }
                v.a.push_back(value());
                skip_ws_nl();
                if (peek() == ',') { ++p_; continue; }
                if (peek() == ']') { ++p_; break; }
                err("expected ',' or ']' in array");
            }
            return v;
        }
        if (c == '{') {
            ++p_;
            v.kind = Value::Table;
            skip_ws();
            if (peek() == '}') { ++p_; return v; }
            for (;;) {
                std::vector<std::string> path = key_path();
                expect('=');
                skip_ws();
                Value e = value();
                assign(v, path, std::move(e));
                skip_ws();
                if (peek() == ',') { ++p_; continue; }
                if (peek() == '}') { ++p_; break; }
                err("expected ',' or '}' in inline table");
            }
            return v;
        }
        // bare scalar: boolean, integer or float
        size_t b = p_;
        while (!eof() && (isalnum((unsigned char)peek()) || peek() == '+' || peek() == '-' || peek() == '.' || peek() == '_' || peek() == ':')) ++p_;
        std::string tok = s_.substr(b, p_ - b);
        if (tok.empty()) err("expected a value");
        if (tok == "true" || tok == "false") { v.kind = Value::Bool; v.b = tok == "true"; return v; }
        std::string num;
        for (char ch : tok) if (ch != '_') num += ch;
        const std::string body = (num[0] == '+' || num[0] == '-') ? num.substr(1) : num;
        if (body == "inf" || body == "nan") {
            v.kind = Value::Float;
            v.f = body == "inf" ? HUGE_VAL : NAN;
            if (num[0] == '-') v.f = -v.f;
            return v;
        }
        if (body.size() > 2 && body[0] == '0' && (body[1] == 'x' || body[1] == 'o' || body[1] == 'b')) {
            const int base = body[1] == 'x' ? 16 : body[1] == 'o' ? 8 : 2;
            char* end = nullptr;
            v.kind = Value::Int;
            v.i = (int64_t)strtoull(body.c_str() + 2, &end, base);
            if (*end) err("bad integer " + tok);
            return v;
        }
        const bool is_float = body.find_first_of(".eE") != std::string::npos;
        char* end = nullptr;
        if (is_float) { v.kind = Value::Float; v.f = strtod(num.c_str(), &end); }
        else { v.kind = Value::Int; v.i = strtoll(num.c_str(), &end, 10); }
        if (!end || *end || !isdigit((unsigned char)body[0])) err("unsupported value " + tok);
        return v;
    }
};

// ---- nalgebra restatements (binary32, same association order) ---------------------------------------

struct Mat4 { float m[4][4]; };      // m[row][col]

Mat4 identity() { Mat4 r{}; for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f; return r; }

// Matrix4::from([[f32;4];4]): each inner array is a COLUMN
Mat4 matrix_from_columns(const Value& v, const char* what) {
    if (v.kind != Value::Array || v.a.size() != 4) fail(std::string(what) + ": a transform must be a 4x4 array");
    Mat4 r{};
    for (int c = 0; c < 4; ++c) {
        const Value& col = v.a[c];
        if (col.kind != Value::Array || col.a.size() != 4) fail(std::string(what) + ": a transform must be a 4x4 array");
        for (int k = 0; k < 4; ++k) {
            if (!col.a[k].is_number()) fail(std::string(what) + ": a transform must hold numbers");
            r.m[k][c] = (float)col.a[k].number();
        }
    }
    return r;
}

void vec3_of(const Value* v, const char* what, float out[3]) {
    if (!v || v->kind != Value::Array || v->a.size() != 3) fail(std::string(what) + ": expected an array of 3 numbers");
    for (int k = 0; k < 3; ++k) {
        if (!v->a[k].is_number()) fail(std::string(what) + ": expected an array of 3 numbers");
        out[k] = (float)v->a[k].number();
    }
}

float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
void normalize3(float v[3]) {
    const float n = std::sqrt(dot3(v, v));
    for (int k = 0; k < 3; ++k) v[k] = v[k] / n;
}

// Matrix4::look_at_rh (scene.rs:165)
Mat4 look_at_rh(const float eye[3], const float target[3], const float up[3]) {
    float z[3] = {eye[0] - target[0], eye[1] - target[1], eye[2] - target[2]}, x[3], y[3];
    normalize3(z);
    cross3(up, z, x);
    normalize3(x);
    cross3(z, x, y);
    Mat4 r{};
    for (int k = 0; k < 3; ++k) { r.m[0][k] = x[k]; r.m[1][k] = y[k]; r.m[2][k] = z[k]; }
    r.m[0][3] = -dot3(x, eye); r.m[1][3] = -dot3(y, eye); r.m[2][3] = -dot3(z, eye);
    r.m[3][3] = 1.0f;
    return r;
}

// a * b, accumulated over k = 0..3 in order (column axpy)
Mat4 mat_mul(const Mat4& a, const Mat4& b) {
    Mat4 r{};
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float acc = a.m[i][0] * b.m[0][j];
            for (int k = 1; k < 4; ++k) acc = acc + a.m[i][k] * b.m[k][j];
            r.m[i][j] = acc;
        }
    return r;
}

void mat3_vec(const Mat4& m, const float v[3], float o[3]) {
    for (int i = 0; i < 3; ++i) {
        float r = m.m[i][0] * v[0];
        r = r + m.m[i][1] * v[1];
        o[i] = r + m.m[i][2] * v[2];
    }
}
float row_dot(const Mat4& m, const float v[3]) { return (m.m[3][0] * v[0] + m.m[3][1] * v[1]) + m.m[3][2] * v[2]; }

void transform_point(const Mat4& m, const float p[3], float o[3]) {
    const float n = row_dot(m, p) + m.m[3][3];
    mat3_vec(m, p, o);
    for (int i = 0; i < 3; ++i) o[i] = o[i] + m.m[i][3];
    if (n != 0.0f) for (int i = 0; i < 3; ++i) o[i] = o[i] / n;
}
void transform_vector(const Mat4& m, const float v[3], float o[3]) {
    float w[3] = {v[0], v[1], v[2]};
    const float n = row_dot(m, w);
    if (n != 0.0f) for (int i = 0; i < 3; ++i) w[i] = w[i] / n;
    mat3_vec(m, w, o);
}

// ---- inflate (RFC 1951) + zlib wrapper (RFC 1950) + PNG (8-bit, non-interlaced) -------------------------

class Inflate {
public:
    Inflate(const uint8_t* d, size_t n) : d_(d), n_(n) {}
    std::vector<uint8_t> run() {
        std::vector<uint8_t> out;
        bool last = false;
        while (!last) {
            last = bits(1) != 0;
            const uint32_t type = bits(2);
            if (type == 0) {
                bitbuf_ = 0; bitcnt_ = 0;
                if (pos_ + 4 > n_) fail("PNG: truncated stored block");
                const uint32_t len = d_[pos_] | (d_[pos_ + 1] << 8);
                pos_ += 4;
                if (pos_ + len > n_) fail("PNG: truncated stored block");
                out.insert(out.end(), d_ + pos_, d_ + pos_ + len);
                pos_ += len;
            } else if (type == 1 || type == 2) {
                Huff lit, dist;
                if (type == 1) fixed(lit, dist); else dynamic(lit, dist);
                codes(lit, dist, out);
            } else fail("PNG: bad deflate block type");
        }
        return out;
    }

private:
    struct Huff { uint16_t count[16]; uint16_t symbol[320]; };
    const uint8_t* d_;
    size_t n_, pos_ = 0;
    uint32_t bitbuf_ = 0;
    int bitcnt_ = 0;

    uint32_t bits(int need) {
        uint32_t v = bitbuf_;
        while (bitcnt_ < need) {
            if (pos_ >= n_) fail("PNG: truncated deflate stream");
            v |= (uint32_t)d_[pos_++] << bitcnt_;
            bitcnt_ += 8;
        }
        bitbuf_ = need < 32 ? v >> need : 0;
        bitcnt_ -= need;
        return need < 32 ? v & ((1u << need) - 1u) : v;
    }
    static void build(Huff& h, const uint8_t* len, int n) {
        memset(&h, 0, sizeof(h));
        for (int i = 0; i < n; ++i) h.count[len[i]]++;
        h.count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = offs[i] + h.count[i];
        for (int i = 0; i < n; ++i) if (len[i]) h.symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(const Huff& h) {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= (int)bits(1);
            const int count = h.count[len];
            if (code - count < first) return h.symbol[index + (code - first)];
            index += count;
            first += count;
            first <<= 1;
            code <<= 1;
        }
        fail("PNG: bad Huffman code");
    }
    static void fixed(Huff& lit, Huff& dist) {
        uint8_t l[288];
        int i = 0;
        for (; i < 144; ++i) l[i] = 8;
        for (; i < 256; ++i) l[i] = 9;
        for (; i < 280; ++i) l[i] = 7;
        for (; i < 288; ++i) l[i] = 8;
        build(lit, l, 288);
        uint8_t d[30];
        for (i = 0; i < 30; ++i) d[i] = 5;
        build(dist, d, 30);
    }
    void dynamic(Huff& lit, Huff& dist) {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
        if (nlen > 286 || ndist > 30) fail("PNG: bad dynamic Huffman header");
        uint8_t l[320] = {0};
        for (int i = 0; i < ncode; ++i) l[order[i]] = (uint8_t)bits(3);
        Huff lencode;
        build(lencode, l, 19);
        memset(l, 0, sizeof(l));
        int idx = 0;
        while (idx < nlen + ndist) {
            int sym = decode(lencode);
            if (sym < 16) l[idx++] = (uint8_t)sym;
            else {
                int rep, val = 0;
                if (sym == 16) { if (idx == 0) fail("PNG: bad length repeat"); val = l[idx - 1]; rep = 3 + (int)bits(2); }
                else if (sym == 17) rep = 3 + (int)bits(3);
                else rep = 11 + (int)bits(7);
                if (idx + rep > nlen + ndist) fail("PNG: bad length repeat");
                while (rep--) l[idx++] = (uint8_t)val;
            }
        }
        build(lit, l, nlen);
        build(dist, l + nlen, ndist);
    }
    void codes(const Huff& lit, const Huff& dist, std::vector<uint8_t>& out) {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (;;) {
            int sym = decode(lit);
            if (sym < 256) out.push_back((uint8_t)sym);
            else if (sym == 256) return;
            else {
                sym -= 257;
                if (sym >= 29) fail("PNG: bad length symbol");
                const int len = lbase[sym] + (int)bits(lext[sym]);
                const int ds = decode(dist);
                if (ds >= 30) fail("PNG: bad distance symbol");
                const size_t d = dbase[ds] + bits(dext[ds]);
                if (d > out.size()) fail("PNG: distance too far back");
                for (int k = 0; k < len; ++k) out.push_back(out[out.size() - d]);
            }
        }
    }
};

struct Png { uint32_t w = 0, h = 0; std::vector<uint8_t> rgba; bool has_alpha = false; };   // always 4 channels

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

Png read_png(const std::string& path) {
    const std::string data = read_file(path, true);
    const uint8_t* d = (const uint8_t*)data.data();
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (data.size() < 8 || memcmp(d, sig, 8) != 0) fail(path + ": only PNG textures are supported by this loader");
    size_t pos = 8;
    std::vector<uint8_t> idat, palette, trns;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    while (pos + 12 <= data.size()) {
        const uint32_t len = be32(d + pos);
        const std::string tag(data, pos + 4, 4);
        if (pos + 12 + (size_t)len > data.size()) fail(path + ": truncated PNG chunk");
        const uint8_t* body = d + pos + 8;
        pos += 12 + (size_t)len;
        if (tag == "IHDR") {
            if (len < 13) fail(path + ": bad IHDR");
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (tag == "PLTE") palette.assign(body, body + len);
        else if (tag == "tRNS") trns.assign(body, body + len);
        else if (tag == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (tag == "IEND") break;
    }
    if (ctype < 0 || w == 0 || h == 0) fail(path + ": PNG without IHDR");
    if (depth != 8 || interlace != 0) fail(path + ": only 8-bit non-interlaced PNG is supported");
    int ch;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break;
                     default: fail(path + ": bad PNG colour type"); }
    if (idat.size() < 6) fail(path + ": PNG without image data");
    std::vector<uint8_t> raw = Inflate(idat.data() + 2, idat.size() - 2).run();      // 2-byte zlib header, Adler-32 ignored
    const size_t stride = (size_t)w * ch;
    if (raw.size() < (stride + 1) * h) fail(path + ": PNG image data too short");
    std::vector<uint8_t> img(stride * h), zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t* line = &raw[(stride + 1) * y + 1];
        uint8_t* cur = &img[stride * y];
        const uint8_t* prev = y ? &img[stride * (y - 1)] : zero.data();
        if (ft > 4) fail(path + ": bad PNG filter type");
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            cur[i] = (uint8_t)((line[i] + pred) & 255);
        }
    }
    Png out;
    out.w = w; out.h = h;
    out.rgba.resize((size_t)w * h * 4);
    out.has_alpha = ctype == 4 || ctype == 6 || (ctype == 3 && !trns.empty());
    for (size_t k = 0; k < (size_t)w * h; ++k) {
        uint8_t r, g, b, a = 255;
        const uint8_t* px = &img[k * ch];
        if (ctype == 0) r = g = b = px[0];
        else if (ctype == 2) { r = px[0]; g = px[1]; b = px[2]; }
        else if (ctype == 4) { r = g = b = px[0]; a = px[1]; }
        else if (ctype == 6) { r = px[0]; g = px[1]; b = px[2]; a = px[3]; }
        else {
            if ((size_t)px[0] * 3 + 2 >= palette.size()) fail(path + ": palette index out of range");
            r = palette[px[0] * 3]; g = palette[px[0] * 3 + 1]; b = palette[px[0] * 3 + 2];
            a = px[0] < trns.size() ? trns[px[0]] : 255;
        }
        out.rgba[k * 4] = r; out.rgba[k * 4 + 1] = g; out.rgba[k * 4 + 2] = b; out.rgba[k * 4 + 3] = a;
    }
    return out;
}

// color.rs:6-12
float to_linear_srgb(float u) { return u <= 0.04045f ? u / 12.92f : std::pow((u + 0.055f) / 1.055f, 2.4f); }

struct HostImage { uint32_t w, h, channels; std::vector<float> texels; };

// Image<Color<SRGB>> (material.rs:39-52)
HostImage image_color(const std::string& path) {
    const Png p = read_png(path);
    HostImage im{p.w, p.h, 3, {}};
    im.texels.resize((size_t)p.w * p.h * 3);
    for (size_t k = 0; k < (size_t)p.w * p.h; ++k)
        for (int c = 0; c < 3; ++c) im.texels[k * 3 + c] = to_linear_srgb((float)p.rgba[k * 4 + c] * (1.0f / 255.0f));
    return im;
}
// Image<f32> (material.rs:54-78): alpha/255 when the file has alpha, else luminance of the decoded colour
HostImage image_scalar(const std::string& path) {
    const Png p = read_png(path);
    HostImage im{p.w, p.h, 1, {}};
    im.texels.resize((size_t)p.w * p.h);
    for (size_t k = 0; k < (size_t)p.w * p.h; ++k) {
        if (p.has_alpha) im.texels[k] = (float)p.rgba[k * 4 + 3] / 255.0f;
        else {
            float lin[3];
            for (int c = 0; c < 3; ++c) lin[c] = to_linear_srgb((float)p.rgba[k * 4 + c] * (1.0f / 255.0f));
            im.texels[k] = (0.21263901f * lin[0] + 0.71516868f * lin[1]) + 0.07219232f * lin[2];
        }
    }
    return im;
}

// ---- OBJ / MTL -------------------------------------------------------------------------------------------

std::vector<std::string> tokens(const std::string& line) {
    std::string body = line.substr(0, line.find('#'));
    std::vector<std::string> t;
    std::istringstream ss(body);
    std::string w;
    while (ss >> w) t.push_back(w);
    return t;
}
std::string rest_of(const std::vector<std::string>& t) {
    std::string s;
    for (size_t k = 1; k < t.size(); ++k) { if (k > 1) s += ' '; s += t[k]; }
    return s;
}
float to_f32(const std::string& s, const std::string& where) {
    char* end = nullptr;
    const float v = strtof(s.c_str(), &end);
    if (end == s.c_str() || *end) fail(where + ": bad number '" + s + "'");
    return v;
}

struct Mtl {
    bool has_kd = false, has_ke = false, has_d = false;
    float kd[3] = {0, 0, 0}, ke[3] = {0, 0, 0}, d = 1.0f;
    std::string map_kd, map_ka, map_d, dir;
};

void parse_mtl(const std::string& path, std::map<std::string, Mtl>& out) {
    std::istringstream in(read_file(path, false));
    std::string line;
    Mtl* cur = nullptr;
    std::map<std::string, Mtl> local;
    while (std::getline(in, line)) {
        const std::vector<std::string> t = tokens(line);
        if (t.empty()) continue;
        if (t[0] == "newmtl") { cur = &local[rest_of(t)]; *cur = Mtl(); cur->dir = dir_of(path); continue; }
        if (!cur) continue;
        if ((t[0] == "Kd" || t[0] == "Ke") && t.size() >= 4) {
            float* dst = t[0] == "Kd" ? cur->kd : cur->ke;
            for (int k = 0; k < 3; ++k) dst[k] = to_f32(t[1 + k], path);
            (t[0] == "Kd" ? cur->has_kd : cur->has_ke) = true;
        } else if (t[0] == "d" && t.size() >= 2) { cur->d = to_f32(t[1], path); cur->has_d = true; }
        else if (t[0] == "map_Kd" && t.size() >= 2) cur->map_kd = t.back();
        else if (t[0] == "map_Ka" && t.size() >= 2) cur->map_ka = t.back();
        else if (t[0] == "map_d" && t.size() >= 2) cur->map_d = t.back();
    }
    for (auto& kv : local) out.emplace(kv.first, kv.second);       // the first library that defines a name wins
}

struct FaceVert { int v, vt, vn; };       // -1 = absent
struct Face { std::vector<FaceVert> verts; int material; };   // index into names, -1 = none

struct Obj {
    std::vector<float> pos, tex, nrm;      // xyz / uv / xyz
    std::vector<Face> faces;
    std::vector<std::string> names, libs;
};

int obj_index(const std::string& s, size_t n, const std::string& where) {
    if (s.empty()) return -1;
    char* end = nullptr;
    const long i = strtol(s.c_str(), &end, 10);
    if (*end || i == 0) fail(where + ": bad index '" + s + "'");
    const long r = i > 0 ? i - 1 : (long)n + i;
    if (r < 0 || r >= (long)n) fail(where + ": index " + s + " out of range");
    return (int)r;
}

Obj parse_obj(const std::string& path) {
    std::istringstream in(read_file(path, false));
    std::string line;
    Obj o;
    int mat = -1;
    while (std::getline(in, line)) {
        const std::vector<std::string> t = tokens(line);
        if (t.empty()) continue;
        if (t[0] == "v" && t.size() >= 4) for (int k = 0; k < 3; ++k) o.pos.push_back(to_f32(t[1 + k], path));
        else if (t[0] == "vt" && t.size() >= 2) { o.tex.push_back(to_f32(t[1], path)); o.tex.push_back(t.size() > 2 ? to_f32(t[2], path) : 0.0f); }
        else if (t[0] == "vn" && t.size() >= 4) for (int k = 0; k < 3; ++k) o.nrm.push_back(to_f32(t[1 + k], path));
        else if (t[0] == "usemtl") {
            const std::string name = rest_of(t);
            mat = -1;
            for (size_t k = 0; k < o.names.size(); ++k) if (o.names[k] == name) mat = (int)k;
            if (mat < 0) { o.names.push_back(name); mat = (int)o.names.size() - 1; }
        } else if (t[0] == "mtllib") o.libs.insert(o.libs.end(), t.begin() + 1, t.end());
        else if (t[0] == "f") {
            Face f;
            f.material = mat;
            for (size_t k = 1; k < t.size(); ++k) {
                std::string part[3];
                size_t b = 0;
                for (int c = 0; c < 3; ++c) {
                    const size_t e = t[k].find('/', b);
                    part[c] = t[k].substr(b, e == std::string::npos ? std::string::npos : e - b);
                    if (e == std::string::npos) break;
                    b = e + 1;
                }
                f.verts.push_back(FaceVert{obj_index(part[0], o.pos.size() / 3, path), obj_index(part[1], o.tex.size() / 2, path),
                                           obj_index(part[2], o.nrm.size() / 3, path)});
                if (f.verts.back().v < 0) fail(path + ": face vertex without a position");
            }
            o.faces.push_back(std::move(f));
        }
    }
    return o;
}

}  // namespace

// ---- the loaded scene behind the C ABI ---------------------------------------------------------------------

struct pt_scene_file {
    std::vector<pt_triangle> triangles;
    std::vector<pt_bvh_node> nodes;
    std::vector<pt_material> materials;
    std::vector<HostImage> image_store;
    std::vector<pt_image> images;
    pt_sky sky{};
    pt_camera camera{};
    uint32_t samples = 0;
    float camera_transform[16] = {};        // column-major
};

namespace {

uint32_t add_image(pt_scene_file& f, HostImage im) {
    f.image_store.push_back(std::move(im));
    return (uint32_t)f.image_store.size() - 1u;
}

// Material::from_mtl (material.rs:206-253)
pt_material material_from_mtl(pt_scene_file& f, const Mtl& m) {
    pt_material out;
    pt_material_default(&out);
    if (!m.map_kd.empty()) { out.diffuse.kind = PT_TEX_NEAREST; out.diffuse.image = add_image(f, image_color(join_path(m.dir, m.map_kd))); }
    else if (m.has_kd) for (int k = 0; k < 3; ++k) out.diffuse.value[k] = m.kd[k];         // Color::srgb(kd): taken as linear
    if (!m.map_ka.empty()) {
        out.emit_kind = PT_EMIT_ABSORBED_D65;
        out.absorption.kind = PT_TEX_NEAREST; out.absorption.image = add_image(f, image_color(join_path(m.dir, m.map_ka)));
    } else if (m.has_ke) {
        out.emit_kind = PT_EMIT_ABSORBED_D65;
        for (int k = 0; k < 3; ++k) out.absorption.value[k] = m.ke[k];
    }
    out.base_dissolve = m.has_d ? m.d : 1.0f;
    if (!m.map_d.empty()) { out.dissolve.kind = PT_TEX_NEAREST; out.dissolve.image = add_image(f, image_scalar(join_path(m.dir, m.map_d))); }
    return out;
}

uint32_t as_u32(const Value& v, const char* what) {
    if (v.kind != Value::Int || v.i < 0 || v.i > 0xFFFFFFFFll) fail(std::string(what) + ": expected a non-negative integer");
    return (uint32_t)v.i;
}
float as_f32(const Value& v, const char* what) {
    if (!v.is_number()) fail(std::string(what) + ": expected a number");
    return (float)v.number();
}

// Scene::load (scene.rs:194-282)
void load(const std::string& path, pt_scene_file& f) {
    const std::string text = read_file(path, false);
    const Value doc = Toml(text).parse();

    // scene.rs:142-152
    f.camera = pt_camera{1024u, 768u, 36.0f, 50.0f};
    f.samples = 128u;
    Mat4 camera_transform = identity();
    if (const Value* cam = doc.get("camera")) {
        if (cam->kind != Value::Table) fail("camera: expected a table");
        if (const Value* r = cam->get("resolution")) {
            if (r->kind != Value::Array || r->a.size() != 2) fail("camera.resolution: expected [width, height]");
            f.camera.width = as_u32(r->a[0], "camera.resolution"); f.camera.height = as_u32(r->a[1], "camera.resolution");
        }
        if (const Value* v = cam->get("sensor_width")) f.camera.sensor_width = as_f32(*v, "camera.sensor_width");
        if (const Value* v = cam->get("focal_length")) f.camera.focal_length = as_f32(*v, "camera.focal_length");
        if (const Value* v = cam->get("samples")) f.samples = as_u32(*v, "camera.samples");
        if (const Value* tr = cam->get("transform")) {
            if (tr->kind == Value::Table) {                     // CameraTransform::LookAt (untagged, tried first)
                float eye[3], target[3], up[3];
                if (!tr->get("eye") || !tr->get("target") || !tr->get("up")) fail("camera.transform: expected {eye, target, up} or a 4x4 array");
                vec3_of(tr->get("eye"), "camera.transform.eye", eye);
                vec3_of(tr->get("target"), "camera.transform.target", target);
                vec3_of(tr->get("up"), "camera.transform.up", up);
                camera_transform = look_at_rh(eye, target, up);
            } else camera_transform = matrix_from_columns(*tr, "camera.transform");
        }
    }
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) f.camera_transform[c * 4 + r] = camera_transform.m[r][c];

    f.materials.resize(1);
    pt_material_default(&f.materials[0]);                       // scene.rs:204
    const std::string base = dir_of(path);
    if (const Value* meshes = doc.get("meshes")) {
        if (meshes->kind != Value::Array) fail("meshes: expected an array of tables");
        for (const Value& mesh : meshes->a) {
            if (mesh.kind != Value::Table) fail("meshes: expected an array of tables");
            const Value* mp = mesh.get("path");
            if (!mp || mp->kind != Value::String) fail("meshes: `path` is required");
            const Mat4 mesh_tf = mesh.get("transform") ? matrix_from_columns(*mesh.get("transform"), "meshes.transform") : identity();
            const Mat4 transform = mat_mul(mesh_tf, camera_transform);        // scene.rs:211
            const std::string mesh_path = join_path(base, mp->s);             // path.with_file_name(mesh.path)
            const Obj obj = parse_obj(mesh_path);
            std::map<std::string, Mtl> mtls;
            for (const std::string& lib : obj.libs) parse_mtl(join_path(dir_of(mesh_path), lib), mtls);
            std::vector<float> tpos(obj.pos.size()), tnrm(obj.nrm.size());
            for (size_t k = 0; k + 2 < obj.pos.size(); k += 3) transform_point(transform, &obj.pos[k], &tpos[k]);
            for (size_t k = 0; k + 2 < obj.nrm.size(); k += 3) transform_vector(transform, &obj.nrm[k], &tnrm[k]);
            std::vector<int> material_map(obj.names.size(), -1);
            for (const Face& face : obj.faces) {
                uint32_t material = 0;                                         // None, or ObjMaterial::Ref ("missing material")
                if (face.material >= 0 && mtls.count(obj.names[face.material])) {
                    if (material_map[face.material] < 0) {
                        const pt_material m = material_from_mtl(f, mtls[obj.names[face.material]]);
                        material_map[face.material] = (int)f.materials.size();
                        f.materials.push_back(m);
                    }
                    material = (uint32_t)material_map[face.material];
                }
                const size_t nv = face.verts.size();
                if (nv < 3) continue;                                          // "Ignoring ..." (scene.rs:250-252)
                if (nv > 4) fail("unhandled " + std::to_string(nv) + "-gon");  // scene.rs:262-264
                static const int corners[2][3] = {{0, 1, 2}, {0, 2, 3}};       // scene.rs:253-261
                for (size_t tri = 0; tri < nv - 2; ++tri) {
                    const FaceVert v[3] = {face.verts[corners[tri][0]], face.verts[corners[tri][1]], face.verts[corners[tri][2]]};
                    pt_triangle t;
                    memset(&t, 0, sizeof(t));
                    for (int k = 0; k < 3; ++k) { t.a[k] = tpos[v[0].v * 3 + k]; t.b[k] = tpos[v[1].v * 3 + k]; t.c[k] = tpos[v[2].v * 3 + k]; }
                    const int n_uv = (v[0].vt >= 0) + (v[1].vt >= 0) + (v[2].vt >= 0), n_n = (v[0].vn >= 0) + (v[1].vn >= 0) + (v[2].vn >= 0);
                    if (n_uv != 0 && n_uv != 3) fail("texture coords defined on some vertices but not all");   // scene.rs:43
                    if (n_n != 0 && n_n != 3) fail("normals defined on some vertices but not all");            // scene.rs:53
                    if (n_uv == 3) {
                        t.flags |= PT_TRI_HAS_UV;
                        for (int c = 0; c < 3; ++c) { t.uv[c][0] = obj.tex[v[c].vt * 2]; t.uv[c][1] = obj.tex[v[c].vt * 2 + 1]; }
                    }
                    if (n_n == 3) {
                        t.flags |= PT_TRI_HAS_NORMALS;
                        for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) t.vertex_normals[c][k] = tnrm[v[c].vn * 3 + k];
                    }
                    t.material_index = material;
                    f.triangles.push_back(t);
                }
            }
        }
    }

    const Value* sky = doc.get("sky");
    if (!sky) pt_sky_d65(0.0f, &f.sky);                                        // Sky::default, scene.rs:88-92
    else if (sky->kind != Value::Table) fail("sky: expected a table");
    else if (const Value* power = sky->get("power")) pt_sky_d65(as_f32(*power, "sky.power"), &f.sky);    // untagged: D65 first
    else {
        const Value *t = sky->get("atmospheric_turbidity"), *al = sky->get("ground_albedo");
        if (!t || !al || !sky->get("zenith_direction") || !sky->get("sun_direction"))
            fail("sky: expected {power} or {atmospheric_turbidity, ground_albedo, zenith_direction, sun_direction}");
        float zenith[3], sun[3];
        vec3_of(sky->get("zenith_direction"), "sky.zenith_direction", zenith);
        vec3_of(sky->get("sun_direction"), "sky.sun_direction", sun);
        if (pt_sky_hosek_wilkie(as_f32(*t, "sky.atmospheric_turbidity"), as_f32(*al, "sky.ground_albedo"), zenith, sun,
                                f.camera_transform, &f.sky) != PT_OK)
            fail("sky: the solar elevation must be > 0 (scene.rs:121) and the turbidity within [1, 10]");
    }

    if (!f.triangles.empty()) {
        if (f.triangles.size() >= 0x3FFFFFFFull) fail("more than 2^30-2 triangles");
        f.nodes.resize(2 * f.triangles.size() - 1);
        uint32_t n_nodes = 0;
        const int rc = pt_bvh_build(f.triangles.data(), (uint32_t)f.triangles.size(), f.nodes.data(), &n_nodes);
        if (rc != PT_OK) fail(rc == PT_ERR_BVH_DEGENERATE ? "BVH::build: a child box is empty (bvh.rs:242)" : "BVH::build failed");
        f.nodes.resize(n_nodes);
    }
    f.images.resize(f.image_store.size());
    for (size_t k = 0; k < f.image_store.size(); ++k)
        f.images[k] = pt_image{f.image_store[k].w, f.image_store[k].h, f.image_store[k].channels, f.image_store[k].texels.data()};
}

}  // namespace

extern "C" int pt_scene_file_load(const char* toml_path, pt_scene_file** out) {
    if (!out) return ptc_fail(nullptr, PT_ERR_INVALID, "pt_scene_file_load: out is NULL");
    *out = nullptr;
    if (!toml_path) return ptc_fail(nullptr, PT_ERR_INVALID, "pt_scene_file_load: path is NULL");
    try {
        auto f = std::make_unique<pt_scene_file>();
        load(toml_path, *f);
        *out = f.release();
        return PT_OK;
    } catch (const std::bad_alloc&) {
        return ptc_fail(nullptr, PT_ERR_NOMEM, "pt_scene_file_load: out of host memory");
    } catch (const std::exception& e) {
        return ptc_fail(nullptr, PT_ERR_INVALID, std::string("pt_scene_file_load: ") + e.what());
    }
}

extern "C" void pt_scene_file_free(pt_scene_file* f) { delete f; }

extern "C" int pt_scene_file_get(const pt_scene_file* f, pt_scene_file_info* out) {
    if (!f || !out) return PT_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    out->desc.triangles = f->triangles.data(); out->desc.n_triangles = (uint32_t)f->triangles.size();
    out->desc.nodes = f->nodes.data(); out->desc.n_nodes = (uint32_t)f->nodes.size();
    out->desc.materials = f->materials.data(); out->desc.n_materials = (uint32_t)f->materials.size();
    out->desc.images = f->images.data(); out->desc.n_images = (uint32_t)f->images.size();
    out->desc.sky = f->sky;
    out->camera = f->camera;
    out->samples = f->samples;
    memcpy(out->camera_transform, f->camera_transform, sizeof(out->camera_transform));
    return PT_OK;
}
