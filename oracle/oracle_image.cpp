// oracle_image.cpp — TEST INFRASTRUCTURE (see oracle/README.md). PARITY UNPINNED.
//
// Image output of the reference, restated: Image::saveImageBMP (imageUtil.cu:69-100), createBMPHeaders (:234-257),
// Image::toneMap / gammaCorrect / postProcessImage (:202-232), Image::saveImageCSV_MONO (:123-142) and the packed
// header structs (:19-43). The reference's file is host C++ but includes util.cuh (CUDA headers), so it cannot be
// compiled here; this follows it statement by statement with the float4 algebra of oracle_types.h.
//
// Two places where the reference leaves the result to the platform, and what is fixed here (SURVEY App. D style; the
// product resolves them the same way):
//   * `static_cast<unsigned char>(clamp(c, 0, 1) * 255.0f + 0.5f)` with c = NaN (clamp passes NaN through; toneMap of
//     Inf is NaN too): undefined behaviour in C++. x86 compilers emit cvttss2si (0x80000000 for NaN) and keep the low
//     byte: 0. Defined as 0.
//   * host FMA contraction of toneMap's a*b+c: none (plain x86-64 has no FMA unless asked for) — evaluated as written.
// The reference always writes "renderCSV.csv" into the working directory; here the path is an argument.
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <string>
#include <vector>

#include "oracle.h"

namespace oracle {

#pragma pack(push, 1)
struct BMPFileHeader {          // imageUtil.cu:20-26
    uint16_t bfType;
    uint32_t bfSize;
    uint16_t bfReserved1;
    uint16_t bfReserved2;
    uint32_t bfOffBits;
};
struct BMPInfoHeader {          // imageUtil.cu:28-40
    uint32_t biSize;
    int32_t biWidth;
    int32_t biHeight;
    uint16_t biPlanes;
    uint16_t biBitCount;
    uint32_t biCompression;
    uint32_t biSizeImage;
    int32_t biXPelsPerMeter;
    int32_t biYPelsPerMeter;
    uint32_t biClrUsed;
    uint32_t biClrImportant;
};
#pragma pack(pop)

static float4 clampf4(float4 v, float lo, float hi) {                       // util.cuh:148-155
    return f4(clampf(v.x, lo, hi), clampf(v.y, lo, hi), clampf(v.z, lo, hi), 0.0f);
}

struct Image {                                                              // imageUtil.cuh:6-31
    bool postProcess;
    const int width, height;
    std::vector<float4> pixels;

    Image(int w, int h) : postProcess(true), width(w), height(h), pixels(std::vector<float4>((size_t)w * h)) {}   // :46
    int toIndex(int x, int y) const { return y * width + x; }              // :50-52
    float4 getColor(int x, int y) const { return pixels[toIndex(x, y)]; }  // :64-66

    static float4 toneMap(float4 color) {                                   // :202-211 (ACES fit)
        const float A = 2.51f;
        const float B = 0.03f;
        const float C = 2.43f;
        const float D = 0.59f;
        const float E = 0.14f;
        return clampf4((color * (A * color + f4(B))) / (color * (C * color + f4(D)) + f4(E)), 0.0f, 1.0f);
    }
    static float4 gammaCorrect(float4 c) {                                  // :213-222
        float invGamma = 1.0f / 2.2f;
        return f4(powf(c.x, invGamma), powf(c.y, invGamma), powf(c.z, invGamma), 0.0f);
    }
    std::vector<float4> postProcessImage() const {                          // :224-232
        std::vector<float4> processed;
        for (int i = 0; i < width * height; i++) processed.push_back(gammaCorrect(toneMap(pixels[i])));
        return processed;
    }

    static unsigned char toByte(float c) {                                  // :90-92, NaN defined as 0 (header)
        float v = clampf(c, 0.0f, 1.0f) * 255.0f + 0.5f;
        if (v != v) return 0;
        return static_cast<unsigned char>(v);
    }

    bool saveImageBMP(const std::string& fileName) const {                  // :69-100
        std::vector<float4> data = postProcess ? postProcessImage() : pixels;
        BMPFileHeader fileHeader;
        BMPInfoHeader infoHeader;
        {                                                                   // createBMPHeaders, :234-257
            int rowSize = (3 * width + 3) & (~3);
            int imageSize = rowSize * height;
            fileHeader.bfType = 0x4D42;
            fileHeader.bfSize = sizeof(BMPFileHeader) + sizeof(BMPInfoHeader) + imageSize;
            fileHeader.bfReserved1 = 0;
            fileHeader.bfReserved2 = 0;
            fileHeader.bfOffBits = sizeof(BMPFileHeader) + sizeof(BMPInfoHeader);
            infoHeader.biSize = sizeof(BMPInfoHeader);
            infoHeader.biWidth = width;
            infoHeader.biHeight = height;
            infoHeader.biPlanes = 1;
            infoHeader.biBitCount = 24;
            infoHeader.biCompression = 0;
            infoHeader.biSizeImage = imageSize;
            infoHeader.biXPelsPerMeter = 0;
            infoHeader.biYPelsPerMeter = 0;
            infoHeader.biClrUsed = 0;
            infoHeader.biClrImportant = 0;
        }
        std::ofstream out(fileName, std::ios::binary);
        if (!out) return false;
        out.write((char*)&fileHeader, sizeof(fileHeader));
        out.write((char*)&infoHeader, sizeof(infoHeader));
        int rowSize = (3 * width + 3) & (~3);       // each row padded to a multiple of 4 bytes; the padding bytes stay 0
        float4 c;
        std::vector<unsigned char> row(rowSize);
        for (int y = 0; y < height; y++) {           // row y of the image = row y of the file: y = 0 is the bottom
            for (int x = 0; x < width; x++) {
                c = data[toIndex(x, y)];
                row[x * 3 + 0] = toByte(c.z);
                row[x * 3 + 1] = toByte(c.y);
                row[x * 3 + 2] = toByte(c.x);
            }
            out.write(reinterpret_cast<char*>(row.data()), rowSize);
        }
        out.close();
        return true;
    }

    bool saveImageCSV_MONO(const std::string& fileName, int choice) const { // :123-142
        std::ofstream csvOut(fileName);
        if (!csvOut) return false;
        csvOut << std::scientific << std::setprecision(3);
        float4 c;
        for (int y = 0; y < height; y++) {
            for (int x = 0; x < width; x++) {
                c = getColor(x, y);
                csvOut << (choice == 0 ? c.x : (choice == 1 ? c.y : (choice == 2 ? c.z : c.w)));     // getFloat4Component
                if (x < width - 1) csvOut << ",";
            }
            csvOut << "\n";
        }
        csvOut.close();
        return true;
    }
};

}  // namespace oracle

extern "C" {

// rgba: w*h float4, y = 0 bottom (what finalise leaves in `out_colors`, main.cu:854-870).
int oracle_save_bmp(const char* path, const float* rgba, int w, int h, int post_process) {
    oracle::Image img(w, h);
    std::memcpy(img.pixels.data(), rgba, (size_t)w * h * 16);
    img.postProcess = post_process != 0;
    return img.saveImageBMP(path) ? 0 : -1;
}
int oracle_save_csv_mono(const char* path, const float* rgba, int w, int h, int channel) {
    oracle::Image img(w, h);
    std::memcpy(img.pixels.data(), rgba, (size_t)w * h * 16);
    return img.saveImageCSV_MONO(path, channel) ? 0 : -1;
}
void oracle_tonemap_gamma(const float* rgba, int n, float* out) {          // gammaCorrect(toneMap(.)) per pixel
    for (int i = 0; i < n; i++) {
        oracle::float4 c = oracle::f4(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]);
        oracle::float4 p = oracle::Image::gammaCorrect(oracle::Image::toneMap(c));
        out[4 * i] = p.x; out[4 * i + 1] = p.y; out[4 * i + 2] = p.z; out[4 * i + 3] = p.w;
    }
}

}  // extern "C"
