/*
 * mfcv.h -- just enough of OpenCV's cv::Mat / Eigen::MatrixXi for the host half of the reference's
 * MfSegmentation::performSegmentation (Core/Segmentation/MfSegmentation.cpp:220-522 of martinruenz/maskfusion) to compile with plain
 * g++ and run.  TEST INFRASTRUCTURE ONLY (same rule as oracle/mf_oracle.h, oracle/ref_shim/, oracle/glsl_shim/).
 *
 * oracle/build_seg.py cuts that statement range out of the reference file in memory (never copied into the repository) and wraps it
 * in a member function of the Harness struct of mfseg_api.cpp, whose members carry the names the slice uses.  What this pins: the
 * label-propagation LOGIC -- ignore map, the five edge-growing sweeps, component x mask and component x model votes, the 65 % / 60 %
 * / 5 % rules, the new-model rule, the bounding-box quirk -- runs from the reference's text.  What it does not: the three OpenCV
 * primitives underneath (connectedComponentsWithStats(4), morphologyEx(MORPH_CLOSE, ellipse), threshold), which are the oracle's
 * restatements (OpenCV is not installed here), called through this header.
 */
#ifndef MFCV_H_
#define MFCV_H_

#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <limits>
#include <list>
#include <memory>
#include <vector>

extern "C" int mfo_connected_components4(const uint8_t* bin, int32_t* labels, int32_t* stats, int max_comp, int W, int H);
extern "C" void mfo_morph_close_ellipse(uint8_t* img, int W, int H, int radius, int iterations);

#define CV_8UC1 0
#define CV_32S 4
#define CV_32SC1 4

namespace cv {
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {} };
struct Size { int width, height; Size(int w, int h) : width(w), height(h) {} };
struct Point { int x, y; Point(int x_, int y_) : x(x_), y(y_) {} };
struct Rect { int x, y, width, height; Rect(int a, int b, int c, int d) : x(a), y(b), width(c), height(d) {} };

/* reference-counted dense matrix, 1 or 4 bytes per element; assignment shares the buffer like cv::Mat */
struct Mat {
    int rows = 0, cols = 0, elem = 1;
    std::shared_ptr<std::vector<uint8_t>> buf;
    uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type, const Scalar& s = Scalar(0)) { create(r, c, type); fill(s.v[0]); }
    void create(int r, int c, int type) {
        rows = r; cols = c; elem = (type == CV_8UC1) ? 1 : 4;
        buf = std::make_shared<std::vector<uint8_t>>((size_t)r * c * elem, 0);
        data = buf->data();
    }
    void fill(double v) {
        if (elem == 1) memset(data, (int)v, total());
        else for (size_t i = 0; i < total(); ++i) reinterpret_cast<int32_t*>(data)[i] = (int32_t)v;
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type, Scalar(0)); }
    size_t total() const { return (size_t)rows * cols; }
    template <class T> T& at(int y, int x) { return reinterpret_cast<T*>(data)[(size_t)y * cols + x]; }
    template <class T> const T& at(int y, int x) const { return reinterpret_cast<const T*>(data)[(size_t)y * cols + x]; }
    template <class T> T& at(size_t i) { return reinterpret_cast<T*>(data)[i]; }
    template <class T> T& at(int i) { return reinterpret_cast<T*>(data)[i]; }
    void copyTo(Mat& dst) const {
        dst.rows = rows; dst.cols = cols; dst.elem = elem;
        dst.buf = std::make_shared<std::vector<uint8_t>>(*buf);
        dst.data = dst.buf->data();
    }
};

enum { THRESH_TOZERO = 3, THRESH_TOZERO_INV = 4, MORPH_CLOSE = 3, MORPH_ELLIPSE = 2 };

/* cv::connectedComponentsWithStats(image, labels, stats, centroids, 4): labels CV_32S, stats rows = {left, top, width, height, area} */
inline int connectedComponentsWithStats(const Mat& img, Mat& labels, Mat& stats, Mat& /*centroids*/, int connectivity) {
    (void)connectivity;   /* the reference asks for 4 */
    if (labels.rows != img.rows || labels.cols != img.cols || labels.elem != 4) labels.create(img.rows, img.cols, CV_32S);
    const int maxc = img.rows * img.cols / 2 + 2;
    std::vector<int32_t> st((size_t)maxc * 5, 0);
    const int n = mfo_connected_components4(img.data, reinterpret_cast<int32_t*>(labels.data), st.data(), maxc, img.cols, img.rows);
    stats.create(n, 5, CV_32S);
    memcpy(stats.data, st.data(), (size_t)n * 5 * sizeof(int32_t));
    return n;
}
inline double threshold(const Mat& src, Mat& dst, double thresh, double /*maxval*/, int type) {
    if (dst.data != src.data) src.copyTo(dst);
    for (size_t i = 0; i < dst.total(); ++i) {
        const uint8_t v = dst.data[i];
        dst.data[i] = (type == THRESH_TOZERO) ? (v > thresh ? v : 0) : (v > thresh ? 0 : v);
    }
    return thresh;
}
/* the structuring element only carries its radius to morphologyEx (the oracle's restatement builds OpenCV's ellipse itself) */
inline Mat getStructuringElement(int /*shape*/, Size ksize, Point /*anchor*/) { return Mat(ksize.height, ksize.width, CV_8UC1, Scalar(1)); }
inline void morphologyEx(const Mat& src, Mat& dst, int /*op = MORPH_CLOSE*/, const Mat& element, Point /*anchor*/, int iterations) {
    if (dst.data != src.data) src.copyTo(dst);
    mfo_morph_close_ellipse(dst.data, dst.cols, dst.rows, (element.cols - 1) / 2, iterations);
}
inline void rectangle(Mat, Rect, Scalar, int) {}
}  // namespace cv

namespace Eigen {
struct MatrixXi {
    int r = 0, c = 0;
    std::vector<int> v;
    static MatrixXi Zero(int rows, int cols) { MatrixXi m; m.r = rows; m.c = cols; m.v.assign((size_t)rows * cols, 0); return m; }
    int& operator()(int i, int j) { return v[(size_t)i * c + j]; }
    struct Row {
        const MatrixXi* m; int i;
        int maxCoeff(int* index) const {   /* first maximum, like Eigen's visitor */
            int best = m->v[(size_t)i * m->c], at = 0;
            for (int j = 1; j < m->c; ++j) if (m->v[(size_t)i * m->c + j] > best) { best = m->v[(size_t)i * m->c + j]; at = j; }
            *index = at;
            return best;
        }
    };
    Row row(int i) const { return Row{this, i}; }
};
}  // namespace Eigen

#define TICK(name)
#define TOCK(name)

#endif /* MFCV_H_ */
