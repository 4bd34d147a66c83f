// ORBmatcher.h — the descriptor-distance / selection part of ORB_SLAM2::ORBmatcher
// (ORB_SLAM2/include/ORBmatcher.h:37-102) on the GPU.
//
// In scope: DescriptorDistance, the best / second-best selection with
// the reference's strict-'<' first-index rule, the TH_LOW / TH_HIGH / ratio acceptance and the
// rotation-consistency histogram (ComputeThreeMaxima).  The eleven projection/BoW search drivers
// stay with the caller for now (§8f "next" rank 2): they gather candidate index lists on the host
// and then need exactly the batched distance + selection calls below.
#ifndef SLAMIT_SHIM_ORBMATCHER_H
#define SLAMIT_SHIM_ORBMATCHER_H

#include <vector>

#ifdef SLAMIT_USE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "cvlite.h"
#endif

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Hamming distance between two 1x32 CV_8U descriptors (ORBmatcher.cc:1651-1667).  One pair per
    // call is a poor fit for a GPU; loops should use DistanceMatrix / BestTwo instead.
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

    // All-pairs distances of two descriptor tables (rows = descriptors), CV_16U-like vector out.
    static bool DistanceMatrix(const cv::Mat& query, const cv::Mat& train, std::vector<unsigned short>& dist);

    // For every query row: index of the nearest train row (first index on ties), its distance and
    // the second-smallest distance (256 / -1 when train is empty) — the loop body of
    // ORBmatcher.cc:1404-1428 for a whole frame at once.
    static bool BestTwo(const cv::Mat& query, const cv::Mat& train, std::vector<int>& bestIdx,
                        std::vector<int>& bestDist, std::vector<int>& secondDist);

    // Brute-force frame-to-frame matching with the acceptance rule of the reference's searches:
    // bestDist <= th (TH_LOW by default) and bestDist < mfNNratio * secondDist, then, when
    // mbCheckOrientation, the three-maxima rotation histogram over (angle1 - angle2)
    // (ORBmatcher.cc:240-250,271-289).  vnMatches12[i] = train index or -1.  Returns the count.
    int SearchBruteForce(const std::vector<cv::KeyPoint>& keys1, const cv::Mat& desc1,
                         const std::vector<cv::KeyPoint>& keys2, const cv::Mat& desc2,
                         std::vector<int>& vnMatches12, int th = -1);

    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

protected:
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM2

#endif
