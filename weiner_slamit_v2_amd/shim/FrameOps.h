// FrameOps.h — the three Frame member functions that sit between the extractor and the guided search, backed by
// libslamit_hip.so: Frame::ComputeImageBounds (ORB_SLAM2/src/Frame.cc:561-590), Frame::UndistortKeyPoints (:529-559)
// and Frame::AssignFeaturesToGrid (:336-357).  Frame itself (a data-model class) stays the caller's: these are
// templates over it, like shim/Optimizer.h, so the reference's Frame.cc can forward its three bodies here:
//
//     void Frame::UndistortKeyPoints()   { ORB_SLAM2::FrameOps::UndistortAndAssign(*this); }   // does both ...
//     void Frame::AssignFeaturesToGrid() {}                                                     // ... in one device call
//     void Frame::ComputeImageBounds(const cv::Mat& im) { ORB_SLAM2::FrameOps::ComputeImageBounds(*this, im); }
//
// Members used on the caller's type (all from include/Frame.h): N, mvKeys, mvKeysUn, mK (3x3 CV_32F), mDistCoef (4x1 or
// 5x1 CV_32F), mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS] (std::vector<std::size_t>), and the statics mnMinX, mnMaxX, mnMinY,
// mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv.
#ifndef SLAMIT_SHIM_FRAMEOPS_H
#define SLAMIT_SHIM_FRAMEOPS_H

#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#ifdef SLAMIT_USE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "cvlite.h"
#endif

#include "../../include/slamit.h"

namespace ORB_SLAM2 {
namespace FrameOps {

inline int& lastStatus() { static int s = 0; return s; }
inline int LastStatus() { return lastStatus(); }

template <class FrameT>
slamit_camera CameraOf(const FrameT& F) {
    slamit_camera c;
    c.fx = F.mK.template at<float>(0, 0); c.fy = F.mK.template at<float>(1, 1);
    c.cx = F.mK.template at<float>(0, 2); c.cy = F.mK.template at<float>(1, 2);
    c.k1 = F.mDistCoef.template at<float>(0, 0); c.k2 = F.mDistCoef.template at<float>(1, 0);
    c.p1 = F.mDistCoef.template at<float>(2, 0); c.p2 = F.mDistCoef.template at<float>(3, 0);
    c.k3 = F.mDistCoef.rows >= 5 ? F.mDistCoef.template at<float>(4, 0) : 0.f;
    return c;
}

// Frame::ComputeImageBounds (Frame.cc:561-590) + the grid constants of the first-frame branch (:113-118)
template <class FrameT>
void ComputeImageBounds(FrameT& F, const cv::Mat& imLeft) {
    if (F.mDistCoef.template at<float>(0, 0) != 0.0) {
        const slamit_camera c = CameraOf(F);
        const float in[8] = {0.f, 0.f, (float)imLeft.cols, 0.f, 0.f, (float)imLeft.rows, (float)imLeft.cols, (float)imLeft.rows};
        float out[8];
        lastStatus() = slamit_undistort_points(0, &c, in, 4, out);
        if (lastStatus() != SLAMIT_OK) return;
        FrameT::mnMinX = std::min(out[0], out[4]); FrameT::mnMaxX = std::max(out[2], out[6]);
        FrameT::mnMinY = std::min(out[1], out[3]); FrameT::mnMaxY = std::max(out[5], out[7]);
    } else {
        FrameT::mnMinX = 0.0f; FrameT::mnMaxX = imLeft.cols;
        FrameT::mnMinY = 0.0f; FrameT::mnMaxY = imLeft.rows;
    }
    FrameT::mfGridElementWidthInv = static_cast<float>(SLAMIT_FRAME_GRID_COLS) / (FrameT::mnMaxX - FrameT::mnMinX);
    FrameT::mfGridElementHeightInv = static_cast<float>(SLAMIT_FRAME_GRID_ROWS) / (FrameT::mnMaxY - FrameT::mnMinY);
}

// Frame::UndistortKeyPoints + Frame::AssignFeaturesToGrid in one device call: fills mvKeysUn and mGrid
template <class FrameT>
void UndistortAndAssign(FrameT& F) {
    const int n = (int)F.mvKeys.size();
    const slamit_camera c = CameraOf(F);
    static_assert(sizeof(cv::KeyPoint) == sizeof(slamit_kp), "cv::KeyPoint must be the 28-byte record");
    F.mvKeysUn.resize(n);
    std::vector<int32_t> start(SLAMIT_FRAME_GRID_CELLS + 1), items((size_t)std::max(n, 1));
    lastStatus() = slamit_frame_finish(0, &c, reinterpret_cast<const slamit_kp*>(F.mvKeys.data()), n, FrameT::mnMinX, FrameT::mnMinY,
                                       FrameT::mfGridElementWidthInv, FrameT::mfGridElementHeightInv,
                                       reinterpret_cast<slamit_kp*>(F.mvKeysUn.data()), start.data(), items.data());
    for (int x = 0; x < SLAMIT_FRAME_GRID_COLS; ++x)
        for (int y = 0; y < SLAMIT_FRAME_GRID_ROWS; ++y) {
            std::vector<std::size_t>& cell = F.mGrid[x][y];
            cell.clear();
            if (lastStatus() != SLAMIT_OK) continue;   // silent failure like the reference: an empty grid
            const int c0 = start[x * SLAMIT_FRAME_GRID_ROWS + y], c1 = start[x * SLAMIT_FRAME_GRID_ROWS + y + 1];
            cell.reserve(c1 - c0);
            for (int j = c0; j < c1; ++j) cell.push_back((std::size_t)items[j]);
        }
}

}  // namespace FrameOps
}  // namespace ORB_SLAM2

#endif
