// mf_labels.hip -- host half of MfSegmentation::performSegmentation (Core/Segmentation/MfSegmentation.cpp:220-522).
//
// The reference runs this stage on the CPU as well (README.md:52 calls it the CPU-bound part): connected components of
// the non-edge mask, five edge-growing sweeps, component x mask and component x model overlap voting, label closing,
// mask -> model assignment with the new-model rule.  OpenCV is replaced by the small routines below (4-connected
// two-pass union-find labelling numbered in raster order like cv::connectedComponentsWithStats; grey-level closing with
// cv::getStructuringElement(MORPH_ELLIPSE)'s element).  A GPU version is listed as "next" in SURVEY.md 8f-2.
#include "mf_labels.h"

#include <math.h>
#include <string.h>
#include <algorithm>

namespace mf {

namespace {

struct CompStats { int left, top, right, bottom, area; };  // inclusive box

int find_root(std::vector<int>& parent, int a) {
    while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; }
    return a;
}

// labels: 0 = background; components numbered 1.. in raster order of their first pixel
int label_components4(const uint8_t* bin, int W, int H, std::vector<int>& labels, std::vector<CompStats>& stats) {
    const int P = W * H;
    labels.assign(P, 0);
    std::vector<int> parent(1, 0);
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const int i = y * W + x;
            if (!bin[i]) continue;
            const int up = (y > 0 && bin[i - W]) ? labels[i - W] : 0;
            const int left = (x > 0 && bin[i - 1]) ? labels[i - 1] : 0;
            if (!up && !left) {
                parent.push_back((int)parent.size());
                labels[i] = (int)parent.size() - 1;
            } else if (up && left) {
                const int ru = find_root(parent, up), rl = find_root(parent, left);
                const int r = std::min(ru, rl);
                parent[ru] = r; parent[rl] = r;
                labels[i] = r;
            } else {
                labels[i] = up ? up : left;
            }
        }
    }
    // provisional roots -> consecutive ids in raster order of first appearance
    std::vector<int> remap(parent.size(), 0);
    int n = 1;
    stats.assign(1, CompStats{0, 0, W - 1, H - 1, 0});
    for (int i = 0; i < P; ++i) {
        if (!labels[i]) { stats[0].area++; continue; }
        const int r = find_root(parent, labels[i]);
        if (!remap[r]) {
            remap[r] = n++;
            stats.push_back(CompStats{W, H, -1, -1, 0});
        }
        const int l = remap[r];
        labels[i] = l;
        CompStats& s = stats[l];
        const int x = i % W, y = i / W;
        s.left = std::min(s.left, x); s.right = std::max(s.right, x);
        s.top = std::min(s.top, y); s.bottom = std::max(s.bottom, y);
        s.area++;
    }
    return n;
}

// cv::getStructuringElement(MORPH_ELLIPSE, (2r+1)^2): per-row half width
void ellipse_spans(int r, std::vector<int>& lo, std::vector<int>& hi) {
    const int ks = 2 * r + 1;
    lo.resize(ks); hi.resize(ks);
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < ks; ++i) {
        const int dy = i - r;
        const int dx = (int)lrint(r * sqrt((r * r - dy * dy) * inv_r2));
        lo[i] = std::max(r - dx, 0);
        hi[i] = std::min(r + dx + 1, ks);
    }
}

template <bool kDilate>
void morph_grey(const std::vector<uint8_t>& in, std::vector<uint8_t>& out, int W, int H, int r) {
    std::vector<int> lo, hi;
    ellipse_spans(r, lo, hi);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int v = kDilate ? 0 : 255;
            for (int i = 0; i < 2 * r + 1; ++i) {
                const int yy = y + i - r;
                if (yy < 0 || yy >= H) continue;  // constant border: neutral element
                for (int j = lo[i]; j < hi[i]; ++j) {
                    const int xx = x + j - r;
                    if (xx < 0 || xx >= W) continue;
                    const int s = in[(size_t)yy * W + xx];
                    v = kDilate ? std::max(v, s) : std::min(v, s);
                }
            }
            out[(size_t)y * W + x] = (uint8_t)v;
        }
}

}  // namespace

void segmentation_host(const SegParams& prm, int W, int H, const uint8_t* binaryIn, const float* depth, const uint8_t* mask,
                       const int32_t* classIDs, int nMasks, const uint8_t* projectedIDs, const std::vector<SegModelInfo>& models,
                       int nextModelID, bool allowNew, std::vector<uint8_t>& ignoreMap, uint8_t* full, SegResult& result) {
    const int total = W * H;
    const int nModels = (int)models.size();
    const size_t minNewMaskPixels = (size_t)(prm.minRelSizeNew * total);
    const size_t maxNewMaskPixels = (size_t)(prm.maxRelSizeNew * total);
    result.hasNewLabel = false;
    result.newClassID = -1;
    if ((int)ignoreMap.size() != total) ignoreMap.assign(total, 0);

    // person / ignore map (MfSegmentation.cpp:221-235)
    std::vector<uint8_t> binary(binaryIn, binaryIn + total);
    if (nMasks) {
        for (int i = 0; i < total; ++i) {
            const bool person = classIDs[mask_id(mask[i], nMasks)] == prm.personClassID;
            ignoreMap[i] = person ? 255 : 0;
            if (person) binary[i] = 0;
        }
    } else {
        for (int i = 0; i < total; ++i)
            if (ignoreMap[i]) binary[i] = 0;
    }

    // connected components (:239)
    std::vector<int> labels;
    std::vector<CompStats> stats;
    const int nComponents = label_components4(binary.data(), W, H, labels, stats);

    // removeEdges (:243-291): five sweeps; a pixel that is edge (0) or in a < 50 px component takes the label of the first
    // 8-neighbour (row-major order) whose depth is within 8 mm and whose component has > 50 px.  Neighbours are read from
    // the previous sweep (the reference reads cvLabelComps while writing the copy r), so a sweep is order independent.
    if (prm.removeEdges) {
        static const int ox[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
        static const int oy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        std::vector<int> next(labels.size());
        for (int it = 0; it < 5; ++it) {
            next = labels;
            for (int y = 1; y < H - 1; ++y)
                for (int x = 1; x < W - 1; ++x) {
                    const int i = y * W + x;
                    const int c = labels[i];
                    if (c != 0 && stats[c].area >= 50) continue;
                    const float d = depth[i];
                    for (int k = 0; k < 8; ++k) {
                        const int j = (y + oy[k]) * W + (x + ox[k]);
                        const int n = labels[j];
                        if (n != 0 && fabsf(depth[j] - d) < 0.008 && stats[n].area > 50) { next[i] = n; break; }
                    }
                }
            labels.swap(next);
        }
    }

    // component x model and component x mask overlaps (:299-318)
    int idToIndex[256];
    for (int k = 0; k < 256; ++k) idToIndex[k] = 0;  // std::map default-constructs to index 0
    for (int m = 0; m < nModels; ++m) idToIndex[models[m].id & 255] = m;
    std::vector<int> compModel((size_t)nComponents * nModels, 0);
    for (int i = 0; i < total; ++i) compModel[(size_t)labels[i] * nModels + idToIndex[projectedIDs[i]]]++;

    std::vector<int> compToMask(nComponents, 0);
    std::vector<int> maskPixels(std::max(nMasks, 1), 0);
    if (nMasks) {
        std::vector<int> compMask((size_t)nComponents * nMasks, 0);
        for (int i = 0; i < total; ++i) compMask[(size_t)labels[i] * nMasks + mask_id(mask[i], nMasks)]++;
        for (int c = 1; c < nComponents; ++c) {
            const int csize = stats[c].area;
            if (csize <= prm.minMappedComponentSize) continue;  // tiny components stay background
            const int t = (int)(0.65f * csize);
            for (int m = 1; m < nMasks; ++m)
                if (compMask[(size_t)c * nMasks + m] > t) {
                    compToMask[c] = m;
                    maskPixels[m] += csize;
                }
        }
    }
    for (int i = 0; i < total; ++i) full[i] = (uint8_t)compToMask[labels[i]];
    for (int i = 0; i < total; ++i)
        if (ignoreMap[i]) full[i] = 255;

    int maskToID[256];
    for (int k = 0; k < 256; ++k) maskToID[k] = 0;
    maskToID[255] = 255;
    if (nMasks) {
        // closing of the label image (:424-426); cv::morphologyEx with iterations == 0 is a copy
        if (prm.morphMaskIterations > 0) {
            std::vector<uint8_t> a(full, full + total), b(total);
            for (int it = 0; it < prm.morphMaskIterations; ++it) { morph_grey<true>(a, b, W, H, prm.morphMaskRadius); a.swap(b); }
            for (int it = 0; it < prm.morphMaskIterations; ++it) { morph_grey<false>(a, b, W, H, prm.morphMaskRadius); a.swap(b); }
            memcpy(full, a.data(), total);
        }
        for (int m = 1; m < nMasks; ++m) maskToID[m] = (classIDs[m] == prm.personClassID) ? 255 : 0;
        // mask x model overlap (:441-447)
        std::vector<unsigned> overlap((size_t)nModels * 256, 0u);
        for (int i = 0; i < total; ++i)
            for (int b = 0; b < nModels; ++b)
                if (projectedIDs[i] == models[b].id) overlap[(size_t)b * 256 + full[i]]++;
        for (int midx = 1; midx < nMasks; ++midx) {
            if (maskToID[midx] == 255) continue;
            int best = 0;
            unsigned bestOverlap = 0;
            for (int j = 1; j < nModels; ++j)
                if (overlap[(size_t)j * 256 + midx] > bestOverlap) { bestOverlap = overlap[(size_t)j * 256 + midx]; best = j; }
            const bool classMatches = models[best].classID == classIDs[midx];
            if (bestOverlap < prm.minMaskModelOverlap * maskPixels[midx]) best = 0;
            if (best != 0 && classMatches) {
                maskToID[midx] = models[best].id;
            } else if (!result.hasNewLabel && allowNew && (size_t)maskPixels[midx] > minNewMaskPixels &&
                       (size_t)maskPixels[midx] < maxNewMaskPixels && best == 0) {
                maskToID[midx] = nextModelID;
                result.hasNewLabel = true;
                result.newClassID = classIDs[midx];
            } else {
                maskToID[midx] = 255;
            }
        }
    }
    for (int i = 0; i < total; ++i) full[i] = (uint8_t)maskToID[full[i]];

    // unassigned components follow the model that covers > 60 % of them in the projection (:500-522)
    for (int c = 1; c < nComponents; ++c) {
        if (compToMask[c] != 0) continue;
        int bestM = 0, ov = compModel[(size_t)c * nModels];
        for (int m = 1; m < nModels; ++m)
            if (compModel[(size_t)c * nModels + m] > ov) { ov = compModel[(size_t)c * nModels + m]; bestM = m; }
        const int modelID = models[bestM].id;
        if (modelID > 0 && ov > 0.6f * stats[c].area) {
            // the reference scans the bounding box with inclusive "+width/+height" bounds; clamped to the image here
            const int x2 = std::min(stats[c].right + 1, W - 1), y2 = std::min(stats[c].bottom + 1, H - 1);
            for (int y = stats[c].top; y <= y2; ++y)
                for (int x = stats[c].left; x <= x2; ++x)
                    if (labels[y * W + x] == c) full[y * W + x] = (uint8_t)modelID;
        }
    }
}

}  // namespace mf
