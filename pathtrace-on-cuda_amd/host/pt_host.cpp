// pt_host.cpp — host half of the C-ABI that needs no GPU: error channel, default
// parameters, tone-map + 8-bit conversion, PNG writer, camera basis.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_api.h"

static thread_local std::string g_err;

void pt_set_error(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" {

const char* pt_last_error(void) { return g_err.c_str(); }
const char* pt_version(void) { return "ptamd 0.1 (gfx950)"; }

void pt_params_default(PtParams* p)
{
    if (!p) return;
    p->passes = 8;            // NUM_MULTI_SAMPLE, include/CudaUtil.cuh:18
    p->spp_per_pass = 1024;   // NUM_SAMPLE, :19
    p->max_bounce = 8;        // MAX_BOUNCE, :15
    p->rr_bounce = 3;         // RUSSIAN_ROULETTE_BOUNCE, :16
    p->rr_floor = 0.5f;       // PROB_STOP_BOUNCE, :17
    p->max_refract = 8;       // `RefractCnt++>8`, :354
    p->first_pass = 0;
    p->rank = 0;
    p->world = 1;
}

// exportImage (srcs/pathtracer.cu:94-112): c = raw / SampleCnt (vec3::operator/=, which
// multiplies by a reciprocal taken in double, include/CudaVector.cuh:151-158), ACESFilm
// (include/CudaUtil.cuh:383-391), ConverToUint8 = (uchar)(v * 255.99f) (include/image.h:5-8).
// ConverToUint8, include/image.h:5-8 (pinned by the real reference: tests/golden/ref_u8.npz)
static inline unsigned char convert_u8(float value) { return (unsigned char)(value * 255.99f); }
int pt_convert_u8(const float* values, int64_t n, uint8_t* out)
{
    if (!values || !out || n < 0) { pt_set_error("pt_convert_u8: bad argument"); return PT_ERR_INVALID; }
    for (int64_t i = 0; i < n; i++) out[i] = convert_u8(values[i]);
    return PT_OK;
}

int pt_tonemap_u8(const float* raw_rgb, int64_t n_pixels, int32_t sample_cnt, uint8_t* rgb8)
{
    if (!raw_rgb || !rgb8 || n_pixels < 0 || sample_cnt < 1) { pt_set_error("pt_tonemap_u8: bad argument"); return PT_ERR_INVALID; }
    const float k = (float)(1.0 / (double)(float)sample_cnt);
    for (int64_t i = 0; i < n_pixels * 3; i++) {
        const float x = raw_rgb[i] * k;
        const float num = x * (2.51f * x + 0.03f);
        const float den = x * (2.43f * x + 0.59f) + 0.14f;
        const float q = num / den;
        const float s = (q > 0.f) ? ((q < 1.f) ? q : 1.f) : 0.f;      // saturate, CudaVector.cuh:296-303
        rgb8[i] = convert_u8(s);
    }
    return PT_OK;
}

// ---- minimal PNG encoder (stored deflate blocks; no external library) -------------------
static uint32_t crc_table[256];
static bool crc_ready = false;
static void crc_init()
{
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
        crc_table[n] = c;
    }
    crc_ready = true;
}
static uint32_t crc32_of(const unsigned char* p, size_t n, uint32_t crc = 0)
{
    if (!crc_ready) crc_init();
    uint32_t c = crc ^ 0xffffffffu;
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}
static void put32(std::vector<unsigned char>& v, uint32_t x)
{
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
static void chunk(std::vector<unsigned char>& out, const char* type, const std::vector<unsigned char>& data)
{
    put32(out, (uint32_t)data.size());
    std::vector<unsigned char> td(type, type + 4);
    td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    put32(out, crc32_of(td.data(), td.size()));
}

// Image::WriteTo (srcs/image.cpp:22-25): row-major, top-down, `channels` bytes per pixel.
int pt_write_png(const char* path, const uint8_t* data, int32_t W, int32_t H, int32_t channels)
{
    if (!path || !data || W < 1 || H < 1 || (channels != 1 && channels != 3 && channels != 4)) { pt_set_error("pt_write_png: bad argument"); return PT_ERR_INVALID; }
    std::vector<unsigned char> raw;
    raw.reserve((size_t)H * ((size_t)W * channels + 1));
    for (int y = 0; y < H; y++) {
        raw.push_back(0);   // filter: none
        raw.insert(raw.end(), data + (size_t)y * W * channels, data + (size_t)(y + 1) * W * channels);
    }
    std::vector<unsigned char> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        size_t n = raw.size() - pos; if (n > 65535) n = 65535;
        const bool last = (pos + n == raw.size());
        z.push_back(last ? 1 : 0);
        z.push_back((unsigned char)(n & 0xff)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xff)); z.push_back((unsigned char)((~n >> 8) & 0xff));
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        pos += n;
        if (last) break;
    }
    put32(z, (b << 16) | a);

    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    put32(ihdr, (uint32_t)W); put32(ihdr, (uint32_t)H);
    ihdr.push_back(8); ihdr.push_back(channels == 1 ? 0 : (channels == 3 ? 2 : 6)); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(png, "IHDR", ihdr);
    chunk(png, "IDAT", z);
    chunk(png, "IEND", {});
    FILE* f = fopen(path, "wb");
    if (!f) { pt_set_error("pt_write_png: cannot open %s", path); return PT_ERR_IO; }
    const size_t w = fwrite(png.data(), 1, png.size(), f);
    fclose(f);
    if (w != png.size()) { pt_set_error("pt_write_png: short write to %s", path); return PT_ERR_IO; }
    return PT_OK;
}

// Camera::SetRotation + GetRight (srcs/camera.cpp:32-66) with glm 0.9.9.8's formulas:
// mod(x,y)=x-y*floor(x/y); normalize(v)=v*(1/sqrt(dot(v,v))); dot = (x+y)+z of the products.
void pt_camera_basis(const float rot_deg[3], float forward[3], float up[3], float right[3])
{
    auto dot3 = [](const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    auto norm3 = [&](float* v) { const float k = 1.0f / std::sqrt(dot3(v, v)); v[0] *= k; v[1] *= k; v[2] *= k; };
    float pitch = rot_deg[1];
    pitch = (pitch < 0.f) ? 0.f : ((pitch > 180.f) ? 180.f : pitch);                 // glm::clamp = min(max(x,lo),hi)
    const float yaw = rot_deg[2] - 360.f * std::floor(rot_deg[2] / 360.f);
    const float degToRad = 0.01745329251994329576923690768489f;
    const float ry = pitch * degToRad, rz = yaw * degToRad;
    float f[3] = {-sinf(ry) * sinf(rz), cosf(ry), -sinf(ry) * cosf(rz)};
    norm3(f);
    float u[3] = {cosf(ry) * sinf(rz), sinf(ry), cosf(ry) * cosf(rz)};
    norm3(u);
    float un[3] = {u[0], u[1], u[2]};
    norm3(un);
    const float d = dot3(f, u);
    float u2[3] = {u[0] - d * un[0], u[1] - d * un[1], u[2] - d * un[2]};
    norm3(u2);
    float r[3] = {f[1] * u2[2] - u2[1] * f[2], f[2] * u2[0] - u2[2] * f[0], f[0] * u2[1] - u2[0] * f[1]};   // glm::cross
    norm3(r);
    memcpy(forward, f, 12); memcpy(up, u2, 12); memcpy(right, r, 12);
}

}  // extern "C"
