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


// ---------------------------------------------------------------------------
// TextureFile::Load (texture.cpp:56-90): ".png" through a PNG decoder to 8-bit RGB (the reference
// calls lodepng::decode(..., LCT_RGB)), ".ppm" as binary P6. Own decoder (zlib inflate + the five
// PNG filters): 8-bit grey / RGB / palette / grey+alpha / RGBA, non-interlaced — what lodepng
// converts losslessly to LCT_RGB 8; anything else fails to load, which the scene treats like a
// missing file (TextureMap(NULL), samples black).
#include "scene_graph.h"

#include <cctype>

namespace rtu {
namespace {

bool read_file(const char* path, std::vector<unsigned char>& out) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (n < 0) { fclose(fp); return false; }
    out.resize((size_t)n);
    bool ok = n == 0 || fread(out.data(), 1, (size_t)n, fp) == (size_t)n;
    fclose(fp);
    return ok;
}
uint32_t be32(const unsigned char* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

bool decode_png(const std::vector<unsigned char>& d, Texture& out) {
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (d.size() < 33 || memcmp(d.data(), sig, 8) != 0) return false;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    size_t off = 8;
    while (off + 12 <= d.size()) {
        uint32_t len = be32(&d[off]);
        const unsigned char* type = &d[off + 4];
        if (off + 12 + (size_t)len > d.size()) return false;
        const unsigned char* data = &d[off + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        off += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || depth != 8 || interlace != 0 || w > 32768 || h > 32768) return false;
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    const size_t stride = (size_t)w * ch;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return false;
    std::vector<unsigned char> img(stride * h);
    for (uint32_t y = 0; y < h; y++) {
        const unsigned char* in = &raw[(stride + 1) * y];
        unsigned char* cur = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        const int f = in[0];
        for (size_t x = 0; x < stride; x++) {
            int a = x >= (size_t)ch ? cur[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0;
            int v = in[1 + x];
            if (f == 1) v += a;
            else if (f == 2) v += b;
            else if (f == 3) v += (a + b) / 2;
            else if (f == 4) {
                int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (f != 0) return false;
            cur[x] = (unsigned char)v;
        }
    }
    out.width = (int)w; out.height = (int)h;
    out.rgb.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        unsigned char* o = &out.rgb[3 * i];
        const unsigned char* p = &img[i * ch];
        if (ctype == 2 || ctype == 6) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
        else if (ctype == 0 || ctype == 4) { o[0] = o[1] = o[2] = p[0]; }
        else {  // palette
            if ((size_t)p[0] * 3 + 2 >= plte.size()) return false;
            o[0] = plte[p[0] * 3]; o[1] = plte[p[0] * 3 + 1]; o[2] = plte[p[0] * 3 + 2];
        }
    }
    return true;
}

bool decode_ppm(const std::vector<unsigned char>& d, Texture& out) {  // LoadPPM, texture.cpp:17-52: binary "P6"
    size_t off = 0;
    auto token = [&](std::string& t) {
        t.clear();
        for (;;) {
            while (off < d.size() && isspace(d[off])) off++;
            if (off < d.size() && d[off] == '#') { while (off < d.size() && d[off] != '\n') off++; continue; }
            break;
        }
        while (off < d.size() && !isspace(d[off])) t.push_back((char)d[off++]);
        return !t.empty();
    };
    std::string t;
    if (!token(t) || t != "P6") return false;
    if (!token(t)) return false;
    int w = atoi(t.c_str());
    if (!token(t)) return false;
    int h = atoi(t.c_str());
    if (!token(t)) return false;  // "255"
    off++;                        // the single whitespace after the header
    if (w <= 0 || h <= 0 || off + (size_t)w * h * 3 > d.size()) return false;
    out.width = w; out.height = h;
    out.rgb.assign(d.begin() + (long)off, d.begin() + (long)(off + (size_t)w * h * 3));
    return true;
}

}  // namespace

bool LoadTextureFile(const char* filename, Texture& out) {
    const size_t len = strlen(filename);
    if (len < 3) return false;
    char ext[4] = {(char)tolower(filename[len - 3]), (char)tolower(filename[len - 2]), (char)tolower(filename[len - 1]), 0};
    std::vector<unsigned char> d;
    if (!strcmp(ext, "png")) return read_file(filename, d) && decode_png(d, out);
    if (!strcmp(ext, "ppm")) return read_file(filename, d) && decode_ppm(d, out);
    return false;
}

}  // namespace rtu
