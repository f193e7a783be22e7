/*
 * mfio_cv.h -- just enough of cv::Mat for the reference's log readers (GUI/Tools/KlgLogReader.cpp, ImageLogReader::loadMaskIDs) to compile
 * with plain g++ and run.  TEST INFRASTRUCTURE ONLY (same rule as oracle/cv_shim/, oracle/ref_shim/, oracle/glsl_shim/): oracle/build_io.py
 * puts this in front of the reference's own text (read where it lies, never copied) and pipes the lot to g++.
 * What this pins: the byte format of a .klg file and of a Mask####.txt descriptor as the reference's parse code reads them (field order and
 * widths, the raw / zlib decision, the depth scale, colour flipping, cv::Rect(b, a, d - b, c - a)).  What it does not: JPEG-compressed
 * colour frames (JPEGLoader needs libjpeg; the stub throws) and cv::imread.
 */
#ifndef MFIO_CV_H_
#define MFIO_CV_H_
#include <stdint.h>
#include <string.h>
#include <memory>
#include <stdexcept>
#include <vector>

#define CV_8UC1 0
#define CV_8UC3 16
#define CV_16UC1 2
#define CV_32FC1 5

namespace cv {
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {} };
struct Rect { int x, y, width, height; Rect(int a = 0, int b = 0, int c = 0, int d = 0) : x(a), y(b), width(c), height(d) {} };

inline int mfio_elem_size(int type) { return type == CV_8UC1 ? 1 : type == CV_8UC3 ? 3 : type == CV_16UC1 ? 2 : 4; }

/* reference-counted dense matrix; assignment shares the buffer like cv::Mat */
struct Mat {
    int rows = 0, cols = 0, mtype = CV_8UC1;
    std::shared_ptr<std::vector<uint8_t>> buf;
    uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type) {
        rows = r; cols = c; mtype = type;
        buf = std::make_shared<std::vector<uint8_t>>((size_t)r * c * mfio_elem_size(type), 0);
        data = buf->data();
    }
    int type() const { return mtype; }
    size_t total() const { return (size_t)rows * cols; }
    void copyTo(Mat& dst) const {
        if (dst.rows != rows || dst.cols != cols || dst.mtype != mtype || !dst.data) dst.create(rows, cols, mtype);
        memcpy(dst.data, data, total() * mfio_elem_size(mtype));
    }
    void setTo(const Scalar& s) {
        const int ch = mtype == CV_8UC3 ? 3 : 1;
        if (mtype == CV_8UC1 || mtype == CV_8UC3)
            for (size_t i = 0; i < total(); ++i)
                for (int c = 0; c < ch; ++c) data[i * ch + c] = (uint8_t)s.v[c];
        else throw std::runtime_error("mfio: setTo on this type is not needed by the readers");
    }
    /* cv::Mat::convertTo(dst, CV_32FC1, alpha) from CV_16UC1: OpenCV's cvtScale16u32f works in float -- (float)src * (float)alpha + 0 */
    void convertTo(Mat& dst, int rtype, double alpha = 1.0, double beta = 0.0) const {
        if (mtype != CV_16UC1 || rtype != CV_32FC1) throw std::runtime_error("mfio: only CV_16UC1 -> CV_32FC1 is needed by the readers");
        if (dst.rows != rows || dst.cols != cols || dst.mtype != rtype || !dst.data) dst.create(rows, cols, rtype);
        const uint16_t* s = reinterpret_cast<const uint16_t*>(data);
        float* d = reinterpret_cast<float*>(dst.data);
        const float a = (float)alpha, b = (float)beta;
        for (size_t i = 0; i < total(); ++i) d[i] = (float)s[i] * a + b;
    }
};
}  // namespace cv
#endif
