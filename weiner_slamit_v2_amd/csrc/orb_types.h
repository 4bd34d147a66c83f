// orb_types.h — host/device shared descriptors of the ORB pipeline's HBM layout.
#ifndef SLAMIT_ORB_TYPES_H
#define SLAMIT_ORB_TYPES_H

#include <stddef.h>
#include <stdint.h>

#define ORB_MAX_LEVELS 16
#define ORB_MAX_ROOTS 8          // octree root nodes = round(width/height) of the detection box
#define ORB_MIN_BORDER 16        // EDGE_THRESHOLD - 3   (ORBextractor.cc:789)
#define ORB_CELL_MAX 59          // wCell, hCell < 60 because nCols = floor(width/30)
#define ORB_TILE_MAX (ORB_CELL_MAX + 6)
#define ORB_CC_PAD 32            // ints between two (frame, level) candidate counters: one 128-byte line each, so the
                                 // FAST waves' atomics do not serialise on one L2 line

// One pyramid level.  Planes of all frames of a batch are stored level-major:
//   plane(level, frame) = pyr + plane_off + frame * plane_bytes, rows of `stride` bytes
// (stride = w rounded up to 64 so every row starts on a 64-byte boundary).  Level 0 lives in
// the caller's buffer (its own stride / frame stride) and has no plane of its own.
struct OrbLevel {
    int32_t w, h;
    int32_t stride;
    int32_t quota;             // mnFeaturesPerLevel[level]
    uint64_t plane_off;        // bytes, in the pyramid buffer (levels >= 1) ...
    uint64_t plane_bytes;      // ... per frame
    uint64_t blur_off;         // bytes, in the blurred-pyramid buffer (all levels)
    uint64_t blur_bytes;
    // FAST cell grid (ORBextractor.cc:797-803)
    int32_t nCols, nRows, wCell, hCell;
    int32_t maxBorderX, maxBorderY;   // w - 16, h - 16
    int32_t cell_base;         // index of this level's first cell in the per-frame cell list
    int32_t ncells;
    int32_t blur_tile_base;    // index of this level's first 64x16 blur tile in the per-frame tile list
    // candidate list of (frame, level): cand + cand_off + frame * cand_frame_stride  (elements)
    uint64_t cand_off;
    int32_t cand_cap;
    // octree roots (ORBextractor.cc:556-576)
    int32_t nIni;
    float hX;
    int32_t rootUL[ORB_MAX_ROOTS], rootUR[ORB_MAX_ROOTS];
    int32_t boxH;              // maxBorderY - minBorderY
    // level keypoints: lkp + kp_off + frame * kp_frame_stride
    int32_t kp_off, kp_cap;    // kp_cap = node capacity = max(quota, 4*nIni) + 4
    float scale;               // mvScaleFactor[level]
    float patch_size;          // (float)(int)(31 * scale)
};

// What fast_cells_kernel needs of every level, passed BY VALUE in the kernel arguments: a wave finds its level
// and geometry with scalar loads of known addresses instead of a chain of dependent loads through a table in HBM.
struct FastLevel {
    int32_t cell_base, nCols, wCell, hCell, maxBorderX, maxBorderY, stride, cand_cap;
    uint64_t plane_off, plane_bytes, cand_off;
};
struct FastTab { FastLevel lv[ORB_MAX_LEVELS]; };

// Fused pyramid: one workgroup builds ALL levels of one image region, each level out of the
// previous one held in LDS.  Per (block, level): the region whose pixels this block stores to HBM
// ("own") and the superset it has to compute because deeper levels read it ("need").
struct PyrBox {
    int16_t ox0, oy0, ox1, oy1;
    int16_t nx0, ny0, nx1, ny1;
};
struct PyrTabs {               // cv::resize coefficient tables of level l (from level l-1), device pointers
    const int32_t* xofs; const int16_t* ialpha;
    const int32_t* yofs; const int16_t* ibeta;
};

struct OrbLevelKp {            // a keypoint in level coordinates, after the octree
    int16_t x, y;
    float response;
    float angle;
    float cs, sn;              // cos / sin of the angle as computeOrbDescriptor needs them (set by the IC kernel)
};

#endif
