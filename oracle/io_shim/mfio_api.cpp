// mfio_api.cpp -- C entry points around the reference's log-reader text (see mfio_cv.h and oracle/build_io.py).  TEST INFRASTRUCTURE ONLY.
// Everything between the MFIO_* markers is replaced in memory by text cut out of /root/reference; nothing of it is stored in this repository.
#include <assert.h>
#include <stdio.h>
#include <stdint.h>
#include <zlib.h>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stack>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "mfio_cv.h"

MFIO_RESOLUTION_H

MFIO_MACROS_H

MFIO_FRAMEDATA_H

namespace pangolin {
inline bool FileExists(const std::string& f) { FILE* p = fopen(f.c_str(), "rb"); if (p) fclose(p); return p != nullptr; }
}
// GUI/Tools/JPEGLoader.h (the reference's libjpeg calling sequence and its channel swap), compiled from its text against the installed
// libjpeg.so.8 through oracle/io_shim/jpeglib.h (the image has the library but not its headers)
extern "C" {
#include "jpeglib.h"
}

MFIO_JPEGLOADER_H

MFIO_LOGREADER_CLASS

MFIO_KLGLOGREADER_CLASS

MFIO_KLGLOGREADER_CPP

// GUI/Tools/ImageLogReader.h declares loadMaskIDs as a const member; the definition below is the reference's
struct ImageLogReader {
    void loadMaskIDs(const std::string& descrFile, std::vector<int>* outClassIds, std::vector<cv::Rect>* outROIs) const;
};

MFIO_LOADMASKIDS

extern "C" {
void* mfio_klg_open(const char* path, int W, int H, int flipColors) {
    try {
        Resolution::setResolution(W, H);
        return new KlgLogReader(path, flipColors != 0);
    } catch (...) { return nullptr; }
}
void mfio_klg_close(void* h) { delete static_cast<KlgLogReader*>(h); }
int mfio_klg_num_frames(void* h) { return static_cast<KlgLogReader*>(h)->getNumFrames(); }
int mfio_klg_has_more(void* h) { return static_cast<KlgLogReader*>(h)->hasMore() ? 1 : 0; }
// getNext() + getFrameData(): the frame MainController::run hands to processFrame.  0 on success, -1 on an exception.
int mfio_klg_next(void* h, int64_t* timestamp, float* depth /*H*W*/, uint8_t* rgb /*H*W*3*/) {
    try {
        KlgLogReader* r = static_cast<KlgLogReader*>(h);
        r->getNext();
        FrameDataPointer f = r->getFrameData();
        *timestamp = f->timestamp;
        memcpy(depth, f->depth.data, f->depth.total() * sizeof(float));
        memcpy(rgb, f->rgb.data, f->rgb.total() * 3);
        return 0;
    } catch (...) { return -1; }
}
// ImageLogReader::loadMaskIDs: ids[0] = 0 (background) then the file's class ids; rois as cv::Rect {x, y, width, height}.
int mfio_load_mask_ids(const char* path, int* ids, int max_ids, int* n_ids, int* rois4, int max_rois, int* n_rois) {
    try {
        ImageLogReader r;
        std::vector<int> c;
        std::vector<cv::Rect> b;
        r.loadMaskIDs(path, &c, &b);
        if ((int)c.size() > max_ids || (int)b.size() > max_rois) return -2;
        *n_ids = (int)c.size(); *n_rois = (int)b.size();
        for (size_t i = 0; i < c.size(); ++i) ids[i] = c[i];
        for (size_t i = 0; i < b.size(); ++i) { rois4[4 * i] = b[i].x; rois4[4 * i + 1] = b[i].y; rois4[4 * i + 2] = b[i].width; rois4[4 * i + 3] = b[i].height; }
        return 0;
    } catch (...) { return -1; }
}
}
