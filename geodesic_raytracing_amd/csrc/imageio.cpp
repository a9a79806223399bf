// imageio.cpp — PNG in/out for the headless renderer (host side, not on the hot path).
//
// Reference counterparts: the screenshot path main.cpp:2762-2808 (read the float4 frame, clamp, linear -> sRGB,
// clamp, 8-bit, PNG through sf::Image) and the background loader graphics_settings.cpp:214-243 (sf::Image from a
// PNG, handed to load_mipped_image).  SFML is not part of the reference checkout; this is a minimal PNG codec on zlib:
// writer = RGBA8, filter 0; reader = 8-bit greyscale / RGB / RGBA / palette, non-interlaced, all five filters.
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/geodesic_hip.h"

extern "C" int gr_internal_fail(int code, const char* msg);

namespace {

void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}

void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    put32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, (uint32_t)crc32(0, out.data() + start, (uInt)(out.size() - start)));
}

// lin_to_srgb_single, cl.cl:326-332 (the host applies the same curve, main.cpp:2797)
float lin_to_srgb(float v) { return v <= 0.0031308f ? v * 12.92f : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f; }
float clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }

uint32_t get32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

extern "C" {

int gr_write_png_rgba8(const char* path, const unsigned char* rgba, int width, int height) {
    if (!path || !rgba || width <= 0 || height <= 0) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "bad image");
    std::vector<uint8_t> raw((size_t)height * ((size_t)width * 4 + 1));
    for (int y = 0; y < height; y++) {
        raw[(size_t)y * (width * 4 + 1)] = 0;   // filter type 0
        memcpy(&raw[(size_t)y * (width * 4 + 1) + 1], rgba + (size_t)y * width * 4, (size_t)width * 4);
    }
    uLongf bound = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(bound);
    if (compress2(z.data(), &bound, raw.data(), (uLong)raw.size(), 6) != Z_OK) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "deflate failed");
    z.resize(bound);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)width);
    put32(ihdr, (uint32_t)height);
    ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});   // 8 bit, RGBA, deflate, adaptive filtering, no interlace
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE* f = fopen(path, "wb");
    if (!f) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, (std::string("cannot write ") + path).c_str());
    fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    return GR_OK;
}

// the screenshot conversion of main.cpp:2791-2800: clamp, linear -> sRGB (all four channels), clamp, * 255 truncated
int gr_frame_to_rgba8(const float* frame_rgba_f32, int width, int height, unsigned char* out_rgba8) {
    if (!frame_rgba_f32 || !out_rgba8) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    size_t n = (size_t)width * height * 4;
    for (size_t i = 0; i < n; i++) {
        float v = clamp01(lin_to_srgb(clamp01(frame_rgba_f32[i])));
        out_rgba8[i] = (unsigned char)(v * 255.f);
    }
    return GR_OK;
}

int gr_write_frame_png(const char* path, const float* frame_rgba_f32, int width, int height) {
    std::vector<unsigned char> px((size_t)width * height * 4);
    int rc = gr_frame_to_rgba8(frame_rgba_f32, width, height, px.data());
    if (rc != GR_OK) return rc;
    return gr_write_png_rgba8(path, px.data(), width, height);
}

// Reads a PNG into RGBA8.  Call with out = NULL to get the size.
int gr_read_png_rgba8(const char* path, int* width, int* height, unsigned char* out, size_t capacity) {
    if (!path || !width || !height) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, (std::string("cannot read ") + path).c_str());
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "not a PNG file");
    uint32_t w = 0, h = 0;
    int depth = 0, colour = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    for (size_t p = 8; p + 12 <= file.size();) {
        uint32_t len = get32(&file[p]);
        if (p + 12 + len > file.size()) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "truncated PNG");
        const uint8_t* type = &file[p + 4];
        const uint8_t* data = &file[p + 8];
        if (!memcmp(type, "IHDR", 4)) { w = get32(data); h = get32(data + 4); depth = data[8]; colour = data[9]; interlace = data[12]; }
        else if (!memcmp(type, "PLTE", 4)) palette.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        p += 12 + len;
    }
    if (!w || !h || depth != 8 || interlace != 0) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "unsupported PNG (need 8-bit, non-interlaced)");
    int channels = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? 1 : colour == 4 ? 2 : colour == 6 ? 4 : 0;
    if (!channels) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "unsupported PNG colour type");
    *width = (int)w;
    *height = (int)h;
    if (!out) return GR_OK;
    if (capacity < (size_t)w * h * 4) return gr_internal_fail(GR_ERROR_BUFFER_TOO_SMALL, "buffer too small");
    size_t stride = (size_t)w * channels;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "PNG inflate failed");
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* line = &raw[y * (stride + 1)];
        int filter = line[0];
        for (size_t x = 0; x < stride; x++) {
            int a = x >= (size_t)channels ? cur[x - channels] : 0, b = prev[x], c = x >= (size_t)channels ? prev[x - channels] : 0;
            int v = line[1 + x];
            switch (filter) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: v += paeth(a, b, c); break;
                default: return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "bad PNG filter");
            }
            cur[x] = (uint8_t)v;
        }
        unsigned char* o = out + (size_t)y * w * 4;
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t* s = &cur[(size_t)x * channels];
            switch (colour) {
                case 0: o[0] = o[1] = o[2] = s[0]; o[3] = 255; break;
                case 2: o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 255; break;
                case 3: {
                    size_t i = s[0];
                    o[0] = i * 3 + 2 < palette.size() ? palette[i * 3] : 0;
                    o[1] = i * 3 + 2 < palette.size() ? palette[i * 3 + 1] : 0;
                    o[2] = i * 3 + 2 < palette.size() ? palette[i * 3 + 2] : 0;
                    o[3] = i < trns.size() ? trns[i] : 255;
                    break;
                }
                case 4: o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; break;
                case 6: o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; break;
            }
            o += 4;
        }
        prev.swap(cur);
    }
    return GR_OK;
}

}  // extern "C"
