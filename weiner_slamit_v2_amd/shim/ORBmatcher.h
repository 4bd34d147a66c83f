// ORBmatcher.h — the descriptor-distance / selection part of ORB_SLAM2::ORBmatcher
// (ORB_SLAM2/include/ORBmatcher.h:37-102) on the GPU.
//
// In scope: DescriptorDistance, the best / second-best selection with
// the reference's strict-'<' first-index rule, the TH_LOW / TH_HIGH / ratio acceptance and the
// rotation-consistency histogram (ComputeThreeMaxima), and — §8f "next" rank 2 — the two tracking-thread
// guided searches, SearchByProjection(Frame&, vector<MapPoint*>&, th) and
// SearchByProjection(CurrentFrame, LastFrame, th, bMono): projection and window sizing are host code
// over the caller's Frame / MapPoint types (templates, like Optimizer.h), the grid window query +
// Hamming best/second + greedy assignment of every map point is ONE slamit_guided_search call.
// Beyond §8f: the vocabulary-node drivers SearchByBoW (KeyFrame/Frame and KeyFrame/KeyFrame) and
// SearchForTriangulation -- the walk over the two DBoW2::FeatureVectors and the rotation histogram are host code,
// the matching loops of every common node are ONE slamit_bow_search call; and the loop-closing Sim3 drivers
// SearchByProjection(pKF, Scw, ...), Fuse(pKF, Scw, ...) and SearchBySim3 over the guided search.
#ifndef SLAMIT_SHIM_ORBMATCHER_H
#define SLAMIT_SHIM_ORBMATCHER_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#include <utility>
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

    // Tracking::SearchLocalPoints' search (ORBmatcher.cc:47-131).  Members used on the caller's types:
    //   Frame    : mvKeysUn, mDescriptors, mvpMapPoints, mvuRight, mvScaleFactors, static mnMinX, mnMinY,
    //              mfGridElementWidthInv, mfGridElementHeightInv
    //   MapPoint : mbTrackInView, isBad(), mnTrackScaleLevel, mTrackViewCos, mTrackProjX, mTrackProjY,
    //              GetDescriptor(), Observations()
    // Mono only: a frame with a stereo keypoint (mvuRight > 0) is refused (returns 0, LastStatus() != 0).
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th = 3);

    // Tracking::TrackWithMotionModel's search (ORBmatcher.cc:1332-1474).  Additional members:
    //   Frame    : N, mTcw, mvbOutlier, mvKeys, fx, fy, cx, cy, mb, static mnMaxX, mnMaxY
    //   MapPoint : GetWorldPos()
    template <class FrameT>
    int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono);

    // LocalMapping::SearchInNeighbors' projection fuse (ORBmatcher.cc:829-979), mono.  Projection, depth / viewing-angle
    // checks and the Replace / AddObservation bookkeeping are host code; the per-point window query with the level and
    // reprojection (chi2 5.99) gates and the nearest-descriptor choice of ALL points is one device call (no keypoint is
    // "taken" here: the choices are independent).  Additional members:
    //   KeyFrame : GetRotation(), GetTranslation(), GetCameraCenter(), fx, fy, cx, cy, IsInImage(u, v), mfLogScaleFactor,
    //              mvScaleFactors, mvInvLevelSigma2, mvKeysUn, mvuRight, mDescriptors, GetMapPoint(idx), AddMapPoint(pMP, idx),
    //              mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv
    //   MapPoint : isBad(), IsInKeyFrame(pKF), GetWorldPos(), GetMaxDistanceInvariance(), GetMinDistanceInvariance(),
    //              GetNormal(), PredictScale(dist, logScaleFactor), GetDescriptor(), Observations(), Replace(p),
    //              AddObservation(pKF, idx)
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT* pKF, const std::vector<MapPointT*>& vpMapPoints, const float th = 3.0);

    // Tracking::Relocalization's search (ORBmatcher.cc:1476-1603): the map points of a candidate keyframe projected into the
    // current frame.  Additional members: KeyFrame : GetMapPointMatches(), mvKeysUn ; Frame : mfLogScaleFactor ;
    // MapPoint : GetMaxDistanceInvariance(), GetMinDistanceInvariance(), PredictScale(dist, logScaleFactor).
    // SetT is anything with count(MapPoint*) (std::set in the reference).
    template <class FrameT, class KeyFrameT, class SetT>
    int SearchByProjection(FrameT& CurrentFrame, KeyFrameT* pKF, const SetT& sAlreadyFound, const float th, const int ORBdist);

    // Monocular initialisation's matcher (ORBmatcher.cc:409-524; called from Tracking::MonocularInitialization).  Frame
    // members as for SearchByProjection.  The matching loop with its matched-distance gate and match take-over is one
    // device call (guided-search mode 1); the rotation histogram and the vbPrevMatched update are host code.
    template <class FrameT>
    int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                int windowSize = 10);

    // Relocalization / loop-detection matcher (ORBmatcher.cc:161-290): map points of a keyframe against the features of a
    // frame that fall into the same vocabulary node.  Members used: KeyFrame : GetMapPointMatches(), mFeatVec
    // (node id -> feature indices, ascending node id: DBoW2::FeatureVector or any std::map-like), mDescriptors, mvKeysUn ;
    // Frame : N, mFeatVec, mDescriptors, mvKeys ; MapPoint : isBad().
    template <class KeyFrameT, class FrameT, class MapPointT>
    int SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches);

    // Loop closing's keyframe-to-keyframe matcher (ORBmatcher.cc:526-657).
    template <class KeyFrameT, class MapPointT>
    int SearchByBoW(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12);

    // LocalMapping::CreateNewMapPoints' matcher (ORBmatcher.cc:659-826), monocular: features without a map point, same
    // vocabulary node, epipolar constraint.  Additional KeyFrame members: N, GetMapPoint(i), GetCameraCenter(),
    // GetRotation(), GetTranslation(), fx, fy, cx, cy, mvScaleFactors, mvLevelSigma2, mvuRight.  A keyframe with a stereo
    // keypoint (mvuRight >= 0) or bOnlyStereo is refused (returns 0, LastStatus() != 0).
    template <class KeyFrameT>
    int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const bool bOnlyStereo);

    // Loop closing's Sim3 drivers (ORBmatcher.cc:293-407, 981-1100, 1102-1330): projection with the similarity transform and
    // every geometric test are host code, the window query + Hamming selection of all map points is one guided-search call
    // per direction.  Additional members: KeyFrame : GetMapPoints() (anything with count(MapPoint*)), IsInImage(u, v) ;
    // MapPoint : GetIndexInKeyFrame(pKF).  cv::Mat products follow cv::gemm for CV_32F (products summed in double).
    template <class KeyFrameT, class MapPointT>
    int SearchByProjection(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints, std::vector<MapPointT*>& vpMatched, int th);
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints, float th, std::vector<MapPointT*>& vpReplacePoint);
    template <class KeyFrameT, class MapPointT>
    int SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, const float& s12, const cv::Mat& R12,
                     const cv::Mat& t12, const float th);

    // Node groups common to two feature vectors, and the device call the three templates above make.
    struct BowGroups {
        std::vector<int32_t> q_ptr, q_idx, c_ptr, c_idx;
        BowGroups() : q_ptr(1, 0), c_ptr(1, 0) {}
        int size() const { return (int)q_ptr.size() - 1; }
    };
    template <class FeatVecT>
    static void CommonNodes(const FeatVecT& fv1, const FeatVecT& fv2, BowGroups& g);
    struct EpipolarGate {          // SearchForTriangulation's per-candidate tests
        float F12[9], ex, ey;
        const std::vector<cv::KeyPoint>* keys1;
        const std::vector<cv::KeyPoint>* keys2;
        const std::vector<float>* scaleFactors;
        const std::vector<float>* levelSigma2;
    };
    static bool BowSearch(const cv::Mat& desc1, const std::vector<uint8_t>& valid1, const cv::Mat& desc2, const std::vector<uint8_t>* valid2,
                          const BowGroups& g, int th, bool thInclusive, float nnratio, const EpipolarGate* gate, std::vector<int>& match12);

    // The device call the templates make.  kp_taken / queries are in the order the reference visits them.
    struct GuidedQueries {
        std::vector<float> uvr;
        std::vector<int32_t> lmin, lmax;
        std::vector<uint8_t> desc, valid, takes;
        void add(float u, float v, float r, int l0, int l1, const cv::Mat& d, bool takesKp) {
            uvr.push_back(u); uvr.push_back(v); uvr.push_back(r);
            lmin.push_back(l0); lmax.push_back(l1);
            const uint8_t* p = d.ptr<uint8_t>(0);
            desc.insert(desc.end(), p, p + 32);
            valid.push_back(1); takes.push_back(takesKp ? 1 : 0);
        }
        int size() const { return (int)lmin.size(); }
    };
    static bool GuidedSearch(const std::vector<cv::KeyPoint>& keysUn, const cv::Mat& descriptors,
                             const std::vector<uint8_t>& kpTaken, float minX, float minY, float invW, float invH,
                             const GuidedQueries& q, int thDist, bool useRatio, float nnratio, std::vector<int>& matchKp,
                             float chi2Gate = 0.f, const std::vector<float>* invLevelSigma2 = nullptr, int mode = 0,
                             std::vector<int>* acceptedKp = nullptr);
    static int LastStatus();

    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

protected:
    static void setStatus(int rc);
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

    float mfNNratio;
    bool mbCheckOrientation;
};

// ---- templates --------------------------------------------------------------------------------

template <class FrameT, class MapPointT>
int ORBmatcher::SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th) {
    const bool bFactor = th != 1.0;
    const int n = (int)F.mvKeysUn.size();
    std::vector<uint8_t> taken(n, 0);
    for (int i = 0; i < n; ++i) {
        if (F.mvuRight[i] > 0) { setStatus(-2 /*SLAMIT_ERR_ARG*/); return 0; }
        if (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) taken[i] = 1;
    }
    GuidedQueries q;
    std::vector<MapPointT*> who;
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
        MapPointT* pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView) continue;
        if (pMP->isBad()) continue;
        const int nPredictedLevel = pMP->mnTrackScaleLevel;
        // RadiusByViewingCos (ORBmatcher.cc:134-140)
        float r = pMP->mTrackViewCos > 0.998 ? 2.5f : 4.0f;
        if (bFactor) r *= th;
        q.add(pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel,
              pMP->GetDescriptor(), pMP->Observations() > 0);
        who.push_back(pMP);
    }
    std::vector<int> matchKp;
    if (!GuidedSearch(F.mvKeysUn, F.mDescriptors, taken, F.mnMinX, F.mnMinY, F.mfGridElementWidthInv, F.mfGridElementHeightInv, q,
                      TH_HIGH, true, mfNNratio, matchKp))
        return 0;
    int nmatches = 0;
    for (size_t k = 0; k < who.size(); ++k)
        if (matchKp[k] >= 0) { F.mvpMapPoints[matchKp[k]] = who[k]; nmatches++; }
    return nmatches;
}


// One row of cv::Mat x3Dc = R * x3Dw + t on CV_32F operands as OpenCV 2.4's cv::gemm evaluates it (modules/core/src/matmul.cpp):
// a 3 x 3 by 3 x 1 product with no transposition flag takes gemm's small-matrix branch -- the dot product in FLOAT, left to
// right, then (float)((double)t0 * alpha + (double)c * beta) with alpha = beta = 1.  (The transposed product -R.t() * t goes
// through the generic path, which accumulates in double.)  PARITY UNPINNED: OpenCV's source is not in the reference tree; this
// follows the published 2.4 code as recalled, and tests/test_shim.py::_gemm_row restates it independently in numpy.
inline float slamit_gemm_row3(const float r0, const float r1, const float r2, const float X, const float Y, const float Z, const float t) {
    const float t0 = r0 * X + r1 * Y + r2 * Z;
    return (float)((double)t0 + (double)t);
}

template <class FrameT>
int ORBmatcher::SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono) {
    const int n = (int)CurrentFrame.mvKeysUn.size();
    std::vector<uint8_t> taken(n, 0);
    for (int i = 0; i < n; ++i) {
        if (CurrentFrame.mvuRight[i] > 0) { setStatus(-2 /*SLAMIT_ERR_ARG*/); return 0; }
        if (CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0) taken[i] = 1;
    }
    float Rcw[3][3], tcw[3], Rlw[3][3], tlw[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) { Rcw[r][c] = CurrentFrame.mTcw.template at<float>(r, c); Rlw[r][c] = LastFrame.mTcw.template at<float>(r, c); }
        tcw[r] = CurrentFrame.mTcw.template at<float>(r, 3); tlw[r] = LastFrame.mTcw.template at<float>(r, 3);
    }
    float twc[3];
    // -Rcw.t() * tcw: cv::gemm's generic path (GEMM_1_T) sums in double and rounds once
    for (int r = 0; r < 3; ++r) twc[r] = (float)((double)-Rcw[0][r] * tcw[0] + (double)-Rcw[1][r] * tcw[1] + (double)-Rcw[2][r] * tcw[2]);
    const float tlc2 = slamit_gemm_row3(Rlw[2][0], Rlw[2][1], Rlw[2][2], twc[0], twc[1], twc[2], tlw[2]);   // row 2 of Rlw * twc + tlw (:1346)
    const bool bForward = tlc2 > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc2 > CurrentFrame.mb && !bMono;

    GuidedQueries q;
    std::vector<int> who;   // index into LastFrame
    for (int i = 0; i < LastFrame.N; i++) {
        auto* pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue;
        const cv::Mat x3Dw = pMP->GetWorldPos();
        const float X = x3Dw.template at<float>(0, 0), Y = x3Dw.template at<float>(1, 0), Z = x3Dw.template at<float>(2, 0);
        const float xc = slamit_gemm_row3(Rcw[0][0], Rcw[0][1], Rcw[0][2], X, Y, Z, tcw[0]);
        const float yc = slamit_gemm_row3(Rcw[1][0], Rcw[1][1], Rcw[1][2], X, Y, Z, tcw[1]);
        const float zc = slamit_gemm_row3(Rcw[2][0], Rcw[2][1], Rcw[2][2], X, Y, Z, tcw[2]);
        const float invzc = 1.0 / zc;
        if (invzc < 0) continue;
        const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        const int nLastOctave = LastFrame.mvKeys[i].octave;
        const float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
        int l0, l1;
        if (bForward) { l0 = nLastOctave; l1 = -1; }
        else if (bBackward) { l0 = 0; l1 = nLastOctave; }
        else { l0 = nLastOctave - 1; l1 = nLastOctave + 1; }
        q.add(u, v, radius, l0, l1, pMP->GetDescriptor(), pMP->Observations() > 0);
        who.push_back(i);
    }
    std::vector<int> matchKp;
    if (!GuidedSearch(CurrentFrame.mvKeysUn, CurrentFrame.mDescriptors, taken, CurrentFrame.mnMinX, CurrentFrame.mnMinY,
                      CurrentFrame.mfGridElementWidthInv, CurrentFrame.mfGridElementHeightInv, q, TH_HIGH, false, mfNNratio, matchKp))
        return 0;
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    for (size_t k = 0; k < who.size(); ++k) {
        const int bestIdx2 = matchKp[k];
        if (bestIdx2 < 0) continue;
        CurrentFrame.mvpMapPoints[bestIdx2] = LastFrame.mvpMapPoints[who[k]];
        nmatches++;
        if (mbCheckOrientation) {
            float rot = LastFrame.mvKeysUn[who[k]].angle - CurrentFrame.mvKeysUn[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdx2);
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    CurrentFrame.mvpMapPoints[rotHist[i][j]] = nullptr;
                    nmatches--;
                }
    }
    return nmatches;
}

template <class FrameT, class KeyFrameT, class SetT>
int ORBmatcher::SearchByProjection(FrameT& CurrentFrame, KeyFrameT* pKF, const SetT& sAlreadyFound, const float th, const int ORBdist) {
    const int n = (int)CurrentFrame.mvKeysUn.size();
    std::vector<uint8_t> taken(n, 0);
    for (int i = 0; i < n; ++i) taken[i] = CurrentFrame.mvpMapPoints[i] ? 1 : 0;   // :1543: any map point blocks the keypoint
    float Rcw[3][3], tcw[3], Ow[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Rcw[r][c] = CurrentFrame.mTcw.template at<float>(r, c);
        tcw[r] = CurrentFrame.mTcw.template at<float>(r, 3);
    }
    for (int r = 0; r < 3; ++r) Ow[r] = (float)((double)-Rcw[0][r] * tcw[0] + (double)-Rcw[1][r] * tcw[1] + (double)-Rcw[2][r] * tcw[2]);
    const auto vpMPs = pKF->GetMapPointMatches();
    GuidedQueries q;
    std::vector<int> who;   // index into vpMPs (= keypoint of pKF)
    for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
        auto* pMP = vpMPs[i];
        if (!pMP) continue;
        if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        const cv::Mat x3Dw = pMP->GetWorldPos();
        const float X = x3Dw.template at<float>(0, 0), Y = x3Dw.template at<float>(1, 0), Z = x3Dw.template at<float>(2, 0);
        const float xc = slamit_gemm_row3(Rcw[0][0], Rcw[0][1], Rcw[0][2], X, Y, Z, tcw[0]);
        const float yc = slamit_gemm_row3(Rcw[1][0], Rcw[1][1], Rcw[1][2], X, Y, Z, tcw[1]);
        const float zc = slamit_gemm_row3(Rcw[2][0], Rcw[2][1], Rcw[2][2], X, Y, Z, tcw[2]);
        const float invzc = 1.0 / zc;
        const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        const float PO[3] = {X - Ow[0], Y - Ow[1], Z - Ow[2]};
        const float dist3D = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, CurrentFrame.mfLogScaleFactor);
        const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
        q.add(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1, pMP->GetDescriptor(), true);
        who.push_back((int)i);
    }
    std::vector<int> matchKp;
    if (!GuidedSearch(CurrentFrame.mvKeysUn, CurrentFrame.mDescriptors, taken, CurrentFrame.mnMinX, CurrentFrame.mnMinY,
                      CurrentFrame.mfGridElementWidthInv, CurrentFrame.mfGridElementHeightInv, q, ORBdist, false, mfNNratio, matchKp))
        return 0;
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    for (size_t k = 0; k < who.size(); ++k) {
        const int bestIdx2 = matchKp[k];
        if (bestIdx2 < 0) continue;
        CurrentFrame.mvpMapPoints[bestIdx2] = vpMPs[who[k]];
        nmatches++;
        if (mbCheckOrientation) {
            float rot = pKF->mvKeysUn[who[k]].angle - CurrentFrame.mvKeysUn[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdx2);
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    CurrentFrame.mvpMapPoints[rotHist[i][j]] = nullptr;
                    nmatches--;
                }
    }
    return nmatches;
}

template <class FeatVecT>
void ORBmatcher::CommonNodes(const FeatVecT& fv1, const FeatVecT& fv2, BowGroups& g) {
    // both are ordered by node id: the reference's lower_bound walk (ORBmatcher.cc:184-278) visits exactly the common ids
    typename FeatVecT::const_iterator a = fv1.begin(), b = fv2.begin();
    while (a != fv1.end() && b != fv2.end()) {
        if (a->first == b->first) {
            for (size_t i = 0; i < a->second.size(); ++i) g.q_idx.push_back((int32_t)a->second[i]);
            for (size_t i = 0; i < b->second.size(); ++i) g.c_idx.push_back((int32_t)b->second[i]);
            g.q_ptr.push_back((int32_t)g.q_idx.size());
            g.c_ptr.push_back((int32_t)g.c_idx.size());
            ++a; ++b;
        } else if (a->first < b->first) ++a;
        else ++b;
    }
}

template <class KeyFrameT, class FrameT, class MapPointT>
int ORBmatcher::SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches) {
    const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPointT*>(F.N, static_cast<MapPointT*>(NULL));
    const int n1 = (int)vpMapPointsKF.size();
    std::vector<uint8_t> valid1(n1, 0);
    for (int i = 0; i < n1; ++i) valid1[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();   // :201-206
    BowGroups g;
    CommonNodes(pKF->mFeatVec, F.mFeatVec, g);
    std::vector<int> match12;
    if (!BowSearch(pKF->mDescriptors, valid1, F.mDescriptors, NULL, g, TH_LOW, true, mfNNratio, NULL, match12)) return 0;
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < n1; ++i) {
        const int bestIdxF = match12[i];
        if (bestIdxF < 0) continue;
        vpMapPointMatches[bestIdxF] = vpMapPointsKF[i];
        if (mbCheckOrientation) {
            float rot = pKF->mvKeysUn[i].angle - F.mvKeys[bestIdxF].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdxF);
        }
        nmatches++;
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                vpMapPointMatches[rotHist[i][j]] = static_cast<MapPointT*>(NULL);
                nmatches--;
            }
        }
    }
    return nmatches;
}

template <class KeyFrameT, class MapPointT>
int ORBmatcher::SearchByBoW(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12) {
    const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
    vpMatches12 = std::vector<MapPointT*>(n1, static_cast<MapPointT*>(NULL));
    std::vector<uint8_t> valid1(n1, 0), valid2(n2, 0);
    for (int i = 0; i < n1; ++i) valid1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();   // :563-567
    for (int i = 0; i < n2; ++i) valid2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();   // :581-587
    BowGroups g;
    CommonNodes(pKF1->mFeatVec, pKF2->mFeatVec, g);
    std::vector<int> match12;
    if (!BowSearch(pKF1->mDescriptors, valid1, pKF2->mDescriptors, &valid2, g, TH_LOW, false, mfNNratio, NULL, match12)) return 0;
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int idx1 = 0; idx1 < n1; ++idx1) {
        const int bestIdx2 = match12[idx1];
        if (bestIdx2 < 0) continue;
        vpMatches12[idx1] = vpMapPoints2[bestIdx2];
        if (mbCheckOrientation) {
            float rot = pKF1->mvKeysUn[idx1].angle - pKF2->mvKeysUn[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(idx1);
        }
        nmatches++;
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                vpMatches12[rotHist[i][j]] = static_cast<MapPointT*>(NULL);
                nmatches--;
            }
        }
    }
    return nmatches;
}

template <class KeyFrameT>
int ORBmatcher::SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                                       const bool bOnlyStereo) {
    vMatchedPairs.clear();
    const int n1 = pKF1->N, n2 = pKF2->N;
    bool stereo = bOnlyStereo;
    for (int i = 0; i < n1 && !stereo; ++i) stereo = pKF1->mvuRight[i] >= 0;
    for (int i = 0; i < n2 && !stereo; ++i) stereo = pKF2->mvuRight[i] >= 0;
    if (stereo) { setStatus(-1); return 0; }   // monocular path only (the reference application is MONOCULAR)
    // epipole of camera 1 in image 2 (:665-673)
    const cv::Mat Cw = pKF1->GetCameraCenter(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    float C2[3];
    for (int r = 0; r < 3; ++r)
        C2[r] = R2w.template at<float>(r, 0) * Cw.template at<float>(0, 0) + R2w.template at<float>(r, 1) * Cw.template at<float>(1, 0) +
                R2w.template at<float>(r, 2) * Cw.template at<float>(2, 0) + t2w.template at<float>(r, 0);
    const float invz = 1.0f / C2[2];
    EpipolarGate gate;
    gate.ex = pKF2->fx * C2[0] * invz + pKF2->cx;
    gate.ey = pKF2->fy * C2[1] * invz + pKF2->cy;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) gate.F12[3 * r + c] = F12.template at<float>(r, c);
    gate.keys1 = &pKF1->mvKeysUn; gate.keys2 = &pKF2->mvKeysUn;
    gate.scaleFactors = &pKF2->mvScaleFactors; gate.levelSigma2 = &pKF2->mvLevelSigma2;
    std::vector<uint8_t> valid1(n1, 0), valid2(n2, 0);
    for (int i = 0; i < n1; ++i) valid1[i] = pKF1->GetMapPoint(i) ? 0 : 1;   // :700-704: skip features that have a MapPoint
    for (int i = 0; i < n2; ++i) valid2[i] = pKF2->GetMapPoint(i) ? 0 : 1;   // :724-728
    BowGroups g;
    CommonNodes(pKF1->mFeatVec, pKF2->mFeatVec, g);
    std::vector<int> vMatches12;
    if (!BowSearch(pKF1->mDescriptors, valid1, pKF2->mDescriptors, &valid2, g, TH_LOW, true, 0.f, &gate, vMatches12)) return 0;
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int idx1 = 0; idx1 < n1; ++idx1) {
        const int bestIdx2 = vMatches12[idx1];
        if (bestIdx2 < 0) continue;
        nmatches++;
        if (mbCheckOrientation) {
            float rot = pKF1->mvKeysUn[idx1].angle - pKF2->mvKeysUn[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(idx1);
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                vMatches12[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    vMatchedPairs.reserve(nmatches);
    for (size_t i = 0, iend = vMatches12.size(); i < iend; i++) {
        if (vMatches12[i] < 0) continue;
        vMatchedPairs.push_back(std::make_pair(i, (size_t)vMatches12[i]));
    }
    return nmatches;
}

// ---- Sim3 drivers ------------------------------------------------------------------------------------------------
namespace sim3detail {
// Scw -> Rcw, tcw, Ow exactly as ORBmatcher.cc:301-306 computes them (cv::Mat::dot and cv::gemm sum in double)
inline void decompose(const cv::Mat& Scw, float R[3][3], float t[3], float O[3]) {
    double d = 0;
    for (int c = 0; c < 3; ++c) d += (double)Scw.at<float>(0, c) * (double)Scw.at<float>(0, c);
    const float scw = (float)sqrt(d);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) R[r][c] = Scw.at<float>(r, c) / scw;
        t[r] = Scw.at<float>(r, 3) / scw;
    }
    for (int r = 0; r < 3; ++r) {
        double a = 0;
        for (int c = 0; c < 3; ++c) a += (double)R[c][r] * (double)t[c];
        O[r] = (float)(-a);
    }
}
inline void apply(const float R[3][3], const float t[3], const float p[3], float out[3]) {   // R * p + t
    for (int r = 0; r < 3; ++r) out[r] = R[r][0] * p[0] + R[r][1] * p[1] + R[r][2] * p[2] + t[r];
}
inline float norm3(const float v[3]) { return (float)sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }   // cv::norm
}  // namespace sim3detail

template <class KeyFrameT, class MapPointT>
int ORBmatcher::SearchByProjection(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints, std::vector<MapPointT*>& vpMatched, int th) {
    const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy;
    float R[3][3], t[3], O[3];
    sim3detail::decompose(Scw, R, t, O);
    const int n = (int)pKF->mvKeysUn.size();
    std::vector<MapPointT*> found(vpMatched.begin(), vpMatched.end());   // spAlreadyFound (:309-310)
    GuidedQueries q;
    std::vector<MapPointT*> who;
    for (int iMP = 0, iendMP = (int)vpPoints.size(); iMP < iendMP; iMP++) {
        MapPointT* pMP = vpPoints[iMP];
        bool seen = false;
        for (size_t k = 0; k < found.size() && !seen; ++k) seen = found[k] == pMP;
        if (pMP->isBad() || seen) continue;
        const cv::Mat p3Dw = pMP->GetWorldPos();
        const float P[3] = {p3Dw.template at<float>(0, 0), p3Dw.template at<float>(1, 0), p3Dw.template at<float>(2, 0)};
        float pc[3];
        sim3detail::apply(R, t, P, pc);
        if (pc[2] < 0.0) continue;
        const float invz = 1 / pc[2];
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const float PO[3] = {P[0] - O[0], P[1] - O[1], P[2] - O[2]};
        const float dist = sim3detail::norm3(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        const cv::Mat Pn = pMP->GetNormal();
        const double dot = (double)PO[0] * Pn.template at<float>(0, 0) + (double)PO[1] * Pn.template at<float>(1, 0) + (double)PO[2] * Pn.template at<float>(2, 0);
        if (dot < 0.5 * dist) continue;
        const int nPredictedLevel = pMP->PredictScale(dist, pKF->mfLogScaleFactor);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        q.add(u, v, radius, nPredictedLevel - 1, nPredictedLevel, pMP->GetDescriptor(), true);   // :378 'if(vpMatched[idx]) continue'
        who.push_back(pMP);
    }
    std::vector<uint8_t> taken((size_t)n, 0);
    for (int i = 0; i < n && i < (int)vpMatched.size(); ++i) taken[i] = vpMatched[i] ? 1 : 0;
    std::vector<int> matchKp;
    if (!GuidedSearch(pKF->mvKeysUn, pKF->mDescriptors, taken, pKF->mnMinX, pKF->mnMinY, pKF->mfGridElementWidthInv,
                      pKF->mfGridElementHeightInv, q, TH_LOW, false, mfNNratio, matchKp))
        return 0;
    int nmatches = 0;
    for (size_t k = 0; k < who.size(); ++k)
        if (matchKp[k] >= 0) { vpMatched[matchKp[k]] = who[k]; nmatches++; }
    return nmatches;
}

template <class KeyFrameT, class MapPointT>
int ORBmatcher::Fuse(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints, float th, std::vector<MapPointT*>& vpReplacePoint) {
    const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy;
    float R[3][3], t[3], O[3];
    sim3detail::decompose(Scw, R, t, O);
    const int n = (int)pKF->mvKeysUn.size();
    const auto spAlreadyFound = pKF->GetMapPoints();
    GuidedQueries q;
    std::vector<int> who;
    const int nPoints = (int)vpPoints.size();
    for (int iMP = 0; iMP < nPoints; iMP++) {
        MapPointT* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        const cv::Mat p3Dw = pMP->GetWorldPos();
        const float P[3] = {p3Dw.template at<float>(0, 0), p3Dw.template at<float>(1, 0), p3Dw.template at<float>(2, 0)};
        float pc[3];
        sim3detail::apply(R, t, P, pc);
        if (pc[2] < 0.0f) continue;
        const float invz = (float)(1.0 / pc[2]);   // :1013 'const float invz = 1.0/...' : a double division
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const float PO[3] = {P[0] - O[0], P[1] - O[1], P[2] - O[2]};
        const float dist3D = sim3detail::norm3(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const cv::Mat Pn = pMP->GetNormal();
        const double dot = (double)PO[0] * Pn.template at<float>(0, 0) + (double)PO[1] * Pn.template at<float>(1, 0) + (double)PO[2] * Pn.template at<float>(2, 0);
        if (dot < 0.5 * dist3D) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF->mfLogScaleFactor);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        q.add(u, v, radius, nPredictedLevel - 1, nPredictedLevel, pMP->GetDescriptor(), false);
        who.push_back(iMP);
    }
    std::vector<int> matchKp;
    const std::vector<uint8_t> none((size_t)n, 0);
    if (!GuidedSearch(pKF->mvKeysUn, pKF->mDescriptors, none, pKF->mnMinX, pKF->mnMinY, pKF->mfGridElementWidthInv,
                      pKF->mfGridElementHeightInv, q, TH_LOW, false, mfNNratio, matchKp))
        return 0;
    int nFused = 0;
    for (size_t k = 0; k < who.size(); ++k) {   // the bookkeeping of :1074-1091, in map-point order
        const int bestIdx = matchKp[k];
        if (bestIdx < 0) continue;
        MapPointT* pMP = vpPoints[who[k]];
        MapPointT* pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[who[k]] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

template <class KeyFrameT, class MapPointT>
int ORBmatcher::SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, const float& s12, const cv::Mat& R12,
                             const cv::Mat& t12, const float th) {
    const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;   // both directions use camera 1's intrinsics (:1105-1108)
    const cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    float Ra[3][3], ta[3], Rb[3][3], tb[3], sR12[3][3], sR21[3][3], t12v[3], t21[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            Ra[r][c] = R1w.template at<float>(r, c); Rb[r][c] = R2w.template at<float>(r, c);
            sR12[r][c] = s12 * R12.template at<float>(r, c);                                   // :1119
            sR21[r][c] = (float)((1.0 / s12) * (double)R12.template at<float>(c, r));          // :1120 (double scale factor)
        }
        ta[r] = t1w.template at<float>(r, 0); tb[r] = t2w.template at<float>(r, 0); t12v[r] = t12.template at<float>(r, 0);
    }
    for (int r = 0; r < 3; ++r) {                                                              // :1121 t21 = -sR21 * t12
        double a = 0;
        for (int c = 0; c < 3; ++c) a += (double)sR21[r][c] * (double)t12v[c];
        t21[r] = (float)(-a);
    }
    const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {
        MapPointT* pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    // one direction: map points of `from` (camera pose Rf, tf), moved into the other camera by (sR, ts), searched in `to`
    struct Dir {
        static bool run(ORBmatcher* self, KeyFrameT* to, const std::vector<MapPointT*>& pts, const std::vector<bool>& done, const float Rf[3][3],
                        const float tf[3], const float sR[3][3], const float ts[3], float fx, float fy, float cx, float cy, float th,
                        std::vector<int>& vnMatch) {
            const int N = (int)pts.size();
            vnMatch.assign(N, -1);
            GuidedQueries q;
            std::vector<int> who;
            for (int i = 0; i < N; i++) {
                MapPointT* pMP = pts[i];
                if (!pMP || done[i]) continue;
                if (pMP->isBad()) continue;
                const cv::Mat p3Dw = pMP->GetWorldPos();
                const float P[3] = {p3Dw.template at<float>(0, 0), p3Dw.template at<float>(1, 0), p3Dw.template at<float>(2, 0)};
                float pa[3], pb[3];
                sim3detail::apply(Rf, tf, P, pa);
                sim3detail::apply(sR, ts, pa, pb);
                if (pb[2] < 0.0) continue;
                const float invz = (float)(1.0 / pb[2]);
                const float x = pb[0] * invz, y = pb[1] * invz;
                const float u = fx * x + cx, v = fy * y + cy;
                if (!to->IsInImage(u, v)) continue;
                const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
                const float dist3D = sim3detail::norm3(pb);
                if (dist3D < minDistance || dist3D > maxDistance) continue;
                const int nPredictedLevel = pMP->PredictScale(dist3D, to->mfLogScaleFactor);
                const float radius = th * to->mvScaleFactors[nPredictedLevel];
                q.add(u, v, radius, nPredictedLevel - 1, nPredictedLevel, pMP->GetDescriptor(), false);
                who.push_back(i);
            }
            std::vector<int> matchKp;
            const std::vector<uint8_t> none(to->mvKeysUn.size(), 0);
            if (!GuidedSearch(to->mvKeysUn, to->mDescriptors, none, to->mnMinX, to->mnMinY, to->mfGridElementWidthInv, to->mfGridElementHeightInv, q,
                              TH_HIGH, false, self->mfNNratio, matchKp))
                return false;
            for (size_t k = 0; k < who.size(); ++k) vnMatch[who[k]] = matchKp[k];
            return true;
        }
    };
    std::vector<int> vnMatch1, vnMatch2;
    if (!Dir::run(this, pKF2, vpMapPoints1, vbAlreadyMatched1, Ra, ta, sR21, t21, fx, fy, cx, cy, th, vnMatch1)) return 0;   // :1146-1223
    if (!Dir::run(this, pKF1, vpMapPoints2, vbAlreadyMatched2, Rb, tb, sR12, t12v, fx, fy, cx, cy, th, vnMatch2)) return 0;  // :1225-1304
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {   // check agreement (:1306-1323)
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; nFound++; }
        }
    }
    return nFound;
}

template <class FrameT>
int ORBmatcher::SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                        int windowSize) {
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);
    GuidedQueries q;
    for (int i1 = 0; i1 < n1; ++i1) {
        q.add(vbPrevMatched[i1].x, vbPrevMatched[i1].y, (float)windowSize, 0, 0, F1.mDescriptors.row(i1), false);
        if (F1.mvKeysUn[i1].octave > 0) q.valid.back() = 0;   // ORBmatcher.cc:426-428: only level-0 keypoints are matched
    }
    std::vector<int> matchKp, acceptedKp;
    const std::vector<uint8_t> none((size_t)n2, 0);
    if (!GuidedSearch(F2.mvKeysUn, F2.mDescriptors, none, F2.mnMinX, F2.mnMinY, F2.mfGridElementWidthInv, F2.mfGridElementHeightInv, q,
                      TH_LOW, false, mfNNratio, matchKp, 0.f, nullptr, 1, &acceptedKp))
        return 0;
    int nmatches = 0;
    for (int i1 = 0; i1 < n1; ++i1) { vnMatches12[i1] = matchKp[i1]; nmatches += matchKp[i1] >= 0; }
    if (mbCheckOrientation) {
        // the reference bins a match when it is made (:467-477) -- also the ones a later query takes over, which stay in
        // their bin (and count for the three maxima) but are skipped by the vnMatches12[idx1] >= 0 test below
        std::vector<int> rotHist[30];
        const float factor = 1.0f / HISTO_LENGTH;
        for (int i1 = 0; i1 < n1; ++i1) {
            if (acceptedKp[i1] < 0) continue;
            float rot = F1.mvKeysUn[i1].angle - F2.mvKeysUn[acceptedKp[i1]].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(i1);
        }
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                const int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < n1; ++i1)   // update prev matched
        if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt;
    return nmatches;
}

template <class KeyFrameT, class MapPointT>
int ORBmatcher::Fuse(KeyFrameT* pKF, const std::vector<MapPointT*>& vpMapPoints, const float th) {
    const cv::Mat Rcw = pKF->GetRotation(), tcw = pKF->GetTranslation(), Ow = pKF->GetCameraCenter();
    float R[3][3], t[3], O[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) R[r][c] = Rcw.template at<float>(r, c);
        t[r] = tcw.template at<float>(r, 0); O[r] = Ow.template at<float>(r, 0);
    }
    const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy;
    const int n = (int)pKF->mvKeysUn.size();
    for (int i = 0; i < n; ++i)
        if (pKF->mvuRight[i] >= 0) { setStatus(-2 /*SLAMIT_ERR_ARG*/); return 0; }   // stereo keypoints: not on this path
    GuidedQueries q;
    std::vector<MapPointT*> who;
    for (size_t i = 0; i < vpMapPoints.size(); i++) {
        MapPointT* pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        const cv::Mat p3Dw = pMP->GetWorldPos();
        const float X = p3Dw.template at<float>(0, 0), Y = p3Dw.template at<float>(1, 0), Z = p3Dw.template at<float>(2, 0);
        const float xc = slamit_gemm_row3(R[0][0], R[0][1], R[0][2], X, Y, Z, t[0]);
        const float yc = slamit_gemm_row3(R[1][0], R[1][1], R[1][2], X, Y, Z, t[1]);
        const float zc = slamit_gemm_row3(R[2][0], R[2][1], R[2][2], X, Y, Z, t[2]);
        if (zc < 0.0f) continue;   // depth must be positive
        const float invz = 1 / zc;
        const float x = xc * invz, y = yc * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const float PO[3] = {X - O[0], Y - O[1], Z - O[2]};
        const float dist3D = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);   // cv::norm: double accumulation
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const cv::Mat Pn = pMP->GetNormal();
        const double dot = (double)PO[0] * Pn.template at<float>(0, 0) + (double)PO[1] * Pn.template at<float>(1, 0) + (double)PO[2] * Pn.template at<float>(2, 0);
        if (dot < 0.5 * dist3D) continue;   // viewing angle below 60 degrees
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF->mfLogScaleFactor);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        q.add(u, v, radius, nPredictedLevel - 1, nPredictedLevel, pMP->GetDescriptor(), false);
        who.push_back(pMP);
    }
    std::vector<int> matchKp;
    const std::vector<uint8_t> none((size_t)n, 0);
    if (!GuidedSearch(pKF->mvKeysUn, pKF->mDescriptors, none, pKF->mnMinX, pKF->mnMinY, pKF->mfGridElementWidthInv,
                      pKF->mfGridElementHeightInv, q, TH_LOW, false, mfNNratio, matchKp, 5.99f, &pKF->mvInvLevelSigma2))
        return 0;
    int nFused = 0;
    for (size_t k = 0; k < who.size(); ++k) {
        const int bestIdx = matchKp[k];
        if (bestIdx < 0) continue;
        MapPointT* pMP = who[k];
        // the reference tests these at the top of each iteration, i.e. AFTER the Replace / AddObservation calls of the
        // earlier map points: re-test here (the device's choices do not depend on them)
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        MapPointT* pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

}  // namespace ORB_SLAM2

#endif
