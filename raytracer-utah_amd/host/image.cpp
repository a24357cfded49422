// image.cpp — output side of the boundary: the RenderImage mirror
// (ExternalLibrary/scene.h:539-656) filled from the device's linear float4
// {r,g,b,z}, plus an 8-bit PNG writer (zlib deflate; the reference uses lodepng,
// scene.h:644-654 — only decoded pixels are comparable, not file bytes).
#include "host_internal.h"

#include <zlib.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

struct RtuImage {
    int width = 0, height = 0;
    std::vector<uint8_t> img;       // Color24[W*H]
    std::vector<float>   zbuffer;   // float[W*H]
    std::vector<uint8_t> zimg;      // empty until computed
    std::atomic<int>     num_rendered{0};
};

namespace {

// Color24::FloatToByte = Clamp(int(r*255)) (cyColor.h:245-246). int(float) of a NaN
// or out-of-range value is undefined in C++; x86 (cvttss2si) yields INT_MIN, which
// Clamp turns into 0 — reproduced explicitly.
inline uint8_t float_to_byte(float r) {
    float v = r * 255;
    int i;
    if (!(v > -2147483904.0f && v < 2147483648.0f)) i = (int)0x80000000;
    else i = (int)v;
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}

void put_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start));
    put_be32(out, crc);
}

}  // namespace

extern "C" {

int rtu_write_png(const char* path, const uint8_t* data, int width, int height, int comp) {
    if (!path || !data || width <= 0 || height <= 0 || (comp != 1 && comp != 3)) return -1;
    std::vector<uint8_t> raw;
    size_t stride = (size_t)width * comp;
    raw.reserve((stride + 1) * height);
    for (int y = 0; y < height; y++) {
        raw.push_back(0);  // filter type None
        raw.insert(raw.end(), data + y * stride, data + (y + 1) * stride);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp_data(clen);
    if (compress2(comp_data.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return -2;
    comp_data.resize(clen);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)width);
    put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8);                      // bit depth
    ihdr.push_back(comp == 3 ? 2 : 0);      // colour type: RGB / grey
    ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", comp_data);
    put_chunk(out, "IEND", {});
    FILE* fp = fopen(path, "wb");
    if (!fp) return -3;
    size_t w = fwrite(out.data(), 1, out.size(), fp);
    fclose(fp);
    return w == out.size() ? 0 : -4;
}

RtuImage* rtu_image_create(int width, int height) {
    if (width <= 0 || height <= 0) return nullptr;
    RtuImage* im = new RtuImage;
    im->width = width;
    im->height = height;
    im->img.assign((size_t)width * height * 3, 0);
    im->zbuffer.assign((size_t)width * height, RTU_BIGFLOAT);
    return im;
}

void rtu_image_free(RtuImage* img) { delete img; }
int rtu_image_width(const RtuImage* img) { return img ? img->width : 0; }
int rtu_image_height(const RtuImage* img) { return img ? img->height : 0; }
uint8_t* rtu_image_pixels(RtuImage* img) { return img ? img->img.data() : nullptr; }
float* rtu_image_zbuffer(RtuImage* img) { return img ? img->zbuffer.data() : nullptr; }
uint8_t* rtu_image_zimage(RtuImage* img) { return (img && !img->zimg.empty()) ? img->zimg.data() : nullptr; }
int rtu_image_num_rendered(const RtuImage* img) { return img ? img->num_rendered.load() : 0; }
int rtu_image_is_done(const RtuImage* img) { return img && img->num_rendered.load() >= img->width * img->height; }

// RenderFunctions.cpp:152-160: gamma pow(double(c), 1/2.2) -> float, Color24, store;
// z straight into the z-buffer (recipe W).
void rtu_image_from_rgbz(RtuImage* img, const float* rgbz, int row0, int nrows) {
    if (!img || !rgbz || row0 < 0 || nrows <= 0 || row0 + nrows > img->height) return;
    const int W = img->width;
    for (int r = 0; r < nrows; r++) {
        const float* src = rgbz + (size_t)r * W * 4;
        uint8_t* dst = img->img.data() + (size_t)(row0 + r) * W * 3;
        float* zdst = img->zbuffer.data() + (size_t)(row0 + r) * W;
        for (int x = 0; x < W; x++) {
            for (int k = 0; k < 3; k++) {
                float g = (float)pow((double)src[4 * x + k], 1 / 2.2);
                dst[3 * x + k] = float_to_byte(g);
            }
            zdst[x] = src[4 * x + 3];
        }
    }
    img->num_rendered.fetch_add(nrows * W);
}

// RenderImage::ComputeZBufferImage (scene.h:590-612)
void rtu_image_compute_zimg(RtuImage* img) {
    if (!img) return;
    size_t size = (size_t)img->width * img->height;
    img->zimg.assign(size, 0);
    float zmin = RTU_BIGFLOAT, zmax = 0;
    for (size_t i = 0; i < size; i++) {
        float z = img->zbuffer[i];
        if (z == RTU_BIGFLOAT) continue;
        if (zmin > z) zmin = z;
        if (zmax < z) zmax = z;
    }
    for (size_t i = 0; i < size; i++) {
        float z = img->zbuffer[i];
        if (z == RTU_BIGFLOAT) img->zimg[i] = 0;
        else {
            float f = (zmax - z) / (zmax - zmin);
            img->zimg[i] = float_to_byte(f);  // int(f*255) clamped to [0,255]
        }
    }
}

int rtu_image_save_png(const RtuImage* img, const char* path) {
    if (!img) return -1;
    return rtu_write_png(path, img->img.data(), img->width, img->height, 3);
}

int rtu_image_save_zpng(const RtuImage* img, const char* path) {
    if (!img || img->zimg.empty()) return -1;
    return rtu_write_png(path, img->zimg.data(), img->width, img->height, 1);
}

}  // extern "C"
