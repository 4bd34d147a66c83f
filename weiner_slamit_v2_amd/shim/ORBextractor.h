// ORBextractor.h — drop-in class surface of ORB_SLAM2::ORBextractor backed by libslamit_hip.so.
//
// Keeps what callers of the reference use (ORB_SLAM2/include/ORBextractor.h:45-85): the five-
// argument constructor, operator()(image, mask, keypoints, descriptors), the level/scale getters
// and the public mvImagePyramid member, so Frame.cc:84-90,360-371 and Tracking.cc:156,162 compile
// against it unchanged.  All work happens on the GPU through the C-ABI of include/slamit.h; there
// is no CPU code path (a missing device makes operator() return no keypoints and sets ok()==false,
// the closest analogue of the reference's silent early return at ORBextractor.cc:1068).
#ifndef SLAMIT_SHIM_ORBEXTRACTOR_H
#define SLAMIT_SHIM_ORBEXTRACTOR_H

#include <vector>

#ifdef SLAMIT_USE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else
#include "cvlite.h"
#endif

struct slamit_orb;

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();

    // mask is ignored, as in the reference
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                    cv::OutputArray descriptors);

    int GetLevels() { return nlevels; }
    float GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Filled after each operator() only when pyramid export is on (stereo matching reads it,
    // Frame.cc:596-703; the monocular path never does, so the default skips the 1.2 MB copy).
    std::vector<cv::Mat> mvImagePyramid;
    void SetPyramidExport(bool on) { exportPyramid = on; }

    // slamit additions
    void SetDevice(int device);          // before the first call; default 0
    bool ok() const { return lastStatus == 0; }
    const char* lastError() const;

private:
    ORBextractor(const ORBextractor&);
    ORBextractor& operator=(const ORBextractor&);
    bool bind(int width, int height);

    int nfeatures;
    double scaleFactor;
    int nlevels, iniThFAST, minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<cv::Mat> mvPaddedPyramid;  // backing store of mvImagePyramid's ROIs

    slamit_orb* handle;
    int boundW, boundH, device, lastStatus;
    bool exportPyramid;
};

}  // namespace ORB_SLAM2

#endif
