// Optimizer.h — ORB_SLAM2::Optimizer::LocalBundleAdjustment backed by libslamit_hip.so.
//
// The reference's function (ORB_SLAM2/src/Optimizer.cc:453-778, declared include/Optimizer.h:45)
// does three things: (1) walks the covisibility graph to pick local keyframes, their map points
// and the fixed keyframes (:456-504), (2) builds a g2o graph and runs the 5 + 10 iteration
// schedule (:507-743), (3) erases outlier observations and writes poses / points back under
// Map::mMutexMapUpdate (:746-777).  Here (1) and (3) are host code written against whatever
// KeyFrame / MapPoint / Map types the caller has (a template, so LocalMapping.cc:84's call
//      Optimizer::LocalBundleAdjustment(mpCurrentKeyFrame, &mbAbortBA, mpMap);
// compiles unchanged against the reference's own classes), and (2) is one slamit_ba_solve call.
//
// Members used on the caller's types, all from the reference's headers:
//   KeyFrame : mnId, mnBALocalForKF, mnBAFixedForKF, GetVectorCovisibleKeyFrames(), isBad(),
//              GetMapPointMatches(), GetPose(), SetPose(), mvKeysUn, mvuRight, mvInvLevelSigma2,
//              fx, fy, cx, cy, EraseMapPointMatch(MapPoint*)
//   MapPoint : mnId, mnBALocalForKF, isBad(), GetObservations(), GetWorldPos(), SetWorldPos(),
//              UpdateNormalAndDepth(), EraseObservation(KeyFrame*)
//   Map      : mMutexMapUpdate
// Stereo observations (mvuRight >= 0, :621-650: EdgeStereoSE3ProjectXYZ with the keyframe's mbf, Huber width sqrt(7.815), gate
// 7.815) ride in the same solve: slamit_ba_problem::edge_ur / kf_bf.  (KeyFrame::mbf is read only when a window has one.)
#ifndef SLAMIT_SHIM_OPTIMIZER_H
#define SLAMIT_SHIM_OPTIMIZER_H

#include <math.h>
#include <stdint.h>

#include <list>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#ifdef SLAMIT_USE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "cvlite.h"
#endif

#include "../../include/slamit.h"

namespace ORB_SLAM2 {

class Optimizer {
public:
    template <class KeyFrameT, class MapT>
    static void LocalBundleAdjustment(KeyFrameT* pKF, bool* pbStopFlag, MapT* pMap);

    // Optimizer::PoseOptimization(Frame*) (Optimizer.cc:239-451) on whatever Frame type the caller has:
    // members used: N, mTcw, mvpMapPoints (elements with GetWorldPos()), mvuRight, mvbOutlier, mvKeysUn,
    // mvInvLevelSigma2, fx, fy, cx, cy, mbf (only when a keypoint has mvuRight >= 0: its stereo edge), SetPose().  Returns the number of inliers like the reference.
    // (The reference holds MapPoint::mGlobalMutex while reading the points; the caller's Frame type is
    //  expected to do the same inside GetWorldPos or around this call.)
    template <class FrameT>
    static int PoseOptimization(FrameT* pFrame);

    // Optimizer::BundleAdjustment / GlobalBundleAdjustemnt (Optimizer.cc:40-238; called from
    // Tracking::CreateInitialMapMonocular and LoopClosing::RunGlobalBundleAdjustment): every keyframe and map point handed
    // in, keyframe id 0 fixed, ONE optimize(nIterations) with Huber delta sqrt(5.99) (or no kernel) -- the same device solve
    // with the schedule (nIterations, 0).  Additional members: KeyFrame : mTcwGBA, mnBAGlobalForKF ; MapPoint : mPosGBA,
    // mnBAGlobalForKF ; Map : GetAllKeyFrames(), GetAllMapPoints().  Stereo observations as in LocalBundleAdjustment (thHuber3D).
    template <class KeyFrameT, class MapPointT>
    static void BundleAdjustment(const std::vector<KeyFrameT*>& vpKFs, const std::vector<MapPointT*>& vpMP, int nIterations = 5,
                                 bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
    template <class MapT>
    static void GlobalBundleAdjustemnt(MapT* pMap, int nIterations = 5, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                       const bool bRobust = true) {
        BundleAdjustment(pMap->GetAllKeyFrames(), pMap->GetAllMapPoints(), nIterations, pbStopFlag, nLoopKF, bRobust);
    }

    // Optimizer::OptimizeSim3 (Optimizer.cc:1046-1247; LoopClosing::ComputeSim3): the validity tests and the camera-frame
    // points on the host, the whole two-stage optimisation in one slamit_sim3_optimize call.  Members used: KeyFrame : mK,
    // GetRotation(), GetTranslation(), GetMapPointMatches(), mvKeysUn, mvInvLevelSigma2 ; MapPoint : isBad(), GetWorldPos(),
    // GetIndexInKeyFrame(pKF).  Sim3T is the caller's g2o::Sim3 (rotation().toRotationMatrix()(r, c), translation()[i],
    // scale(), constructible from (Eigen::Matrix3d, Eigen::Vector3d, double)) or any type for which the two
    // slamit_shim_sim3_{get,set} overloads exist.
    template <class KeyFrameT, class MapPointT, class Sim3T>
    static int OptimizeSim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches1, Sim3T& g2oS12, const float th2, const bool bFixScale);

    // POD form: the window already flattened (what the template above produces).
    static int LocalBundleAdjustmentPOD(const slamit_ba_problem& prob, const volatile bool* stop, slamit_ba_result& res);
    static int SolvePOD(const slamit_ba_problem& prob, const slamit_ba_opts& opts, slamit_ba_result& res);

    static int LastStatus() { return lastStatus(); }
    static void SetDevice(int device) { deviceRef() = device; }

private:
    static int& lastStatus() { static thread_local int s = 0; return s; }   // per calling thread: Tracking, LocalMapping and LoopClosing run concurrently
    static int& deviceRef() { static int d = 0; return d; }
    static slamit_ba*& handleRef() { static slamit_ba* h = 0; return h; }
    static int* capRef() { static int c[3] = {0, 0, 0}; return c; }
    static std::mutex& solveMutex() { static std::mutex m; return m; }
};

inline int Optimizer::SolvePOD(const slamit_ba_problem& prob, const slamit_ba_opts& opts, slamit_ba_result& res) {
    std::lock_guard<std::mutex> guard(solveMutex());  // one LocalMapping thread in the reference; be safe anyway
    int* cap = capRef();
    if (!handleRef() || prob.n_kf > cap[0] || prob.n_pt > cap[1] || prob.n_edge > cap[2]) {
        slamit_ba_destroy(handleRef());
        handleRef() = 0;
        cap[0] = prob.n_kf > 64 ? 2 * prob.n_kf : 64;
        cap[1] = prob.n_pt > 4096 ? 2 * prob.n_pt : 4096;
        cap[2] = prob.n_edge > 65536 ? 2 * prob.n_edge : 65536;
        int rc = slamit_ba_create(cap[0], cap[1], cap[2], 1, deviceRef(), &handleRef());
        if (rc != SLAMIT_OK) { handleRef() = 0; cap[0] = cap[1] = cap[2] = 0; return lastStatus() = rc; }
    }
    return lastStatus() = slamit_ba_solve(handleRef(), &prob, &opts, &res);
}

inline int Optimizer::LocalBundleAdjustmentPOD(const slamit_ba_problem& prob, const volatile bool* stop, slamit_ba_result& res) {
    slamit_ba_opts o;
    o.its_robust = 5;                              // Optimizer.cc:660
    o.its_final = 10;                              // :707
    o.huber_delta = (double)(float)sqrt(5.991);    // :569 (a float in the reference)
    o.chi2_gate = 5.991;                           // :680,723
    o.huber_delta_stereo = (double)(float)sqrt(7.815);   // :570
    o.chi2_gate_stereo = 7.815;                    // :696,740
    o.stop = reinterpret_cast<const volatile uint8_t*>(stop);
    return SolvePOD(prob, o, res);
}

template <class KeyFrameT, class MapPointT>
void Optimizer::BundleAdjustment(const std::vector<KeyFrameT*>& vpKFs, const std::vector<MapPointT*>& vpMP, int nIterations,
                                 bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust) {
    // ---- vertices (Optimizer.cc:69-85, 91-105) and edges (:107-160), flattened ----
    std::vector<KeyFrameT*> kfs;
    std::map<KeyFrameT*, int> kfIndex;
    std::vector<double> pose, intr, pts, uv, invSigma2, ur, bf;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> ekf, ept;
    long unsigned int maxKFid = 0;
    for (size_t i = 0; i < vpKFs.size(); i++) {
        KeyFrameT* pKF = vpKFs[i];
        if (pKF->isBad()) continue;
        kfIndex[pKF] = (int)kfs.size();
        kfs.push_back(pKF);
        cv::Mat T = pKF->GetPose();
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) pose.push_back((double)T.template at<float>(r, c));
        for (int r = 0; r < 3; ++r) pose.push_back((double)T.template at<float>(r, 3));
        fixed.push_back(pKF->mnId == 0);
        intr.push_back(pKF->fx); intr.push_back(pKF->fy); intr.push_back(pKF->cx); intr.push_back(pKF->cy);
        if (pKF->mnId > maxKFid) maxKFid = pKF->mnId;
    }
    std::vector<MapPointT*> mps;          // the points that got at least one edge (vbNotIncludedMP == false)
    bool hasStereo = false;
    for (size_t i = 0; i < vpMP.size(); i++) {
        MapPointT* pMP = vpMP[i];
        if (pMP->isBad()) continue;
        const std::map<KeyFrameT*, size_t> observations = pMP->GetObservations();
        int nEdges = 0;
        const int p = (int)mps.size();
        for (typename std::map<KeyFrameT*, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
            KeyFrameT* pKF = mit->first;
            if (pKF->isBad() || pKF->mnId > maxKFid) continue;
            typename std::map<KeyFrameT*, int>::iterator where = kfIndex.find(pKF);
            if (where == kfIndex.end()) continue;
            const cv::KeyPoint& kpUn = pKF->mvKeysUn[mit->second];
            nEdges++;
            ekf.push_back(where->second); ept.push_back(p);
            uv.push_back(kpUn.pt.x); uv.push_back(kpUn.pt.y);
            invSigma2.push_back(pKF->mvInvLevelSigma2[kpUn.octave]);
            const float kp_ur = pKF->mvuRight[mit->second];          // :112 monocular below zero, :135-160 the stereo edge
            ur.push_back(kp_ur < 0 ? -1.0 : (double)kp_ur);
            hasStereo = hasStereo || !(kp_ur < 0);
        }
        if (nEdges == 0) continue;        // optimizer.removeVertex(vPoint)
        cv::Mat X = pMP->GetWorldPos();
        for (int r = 0; r < 3; ++r) pts.push_back((double)X.template at<float>(r, 0));
        mps.push_back(pMP);
    }
    if (kfs.empty()) return;
    if (hasStereo) for (size_t k = 0; k < kfs.size(); ++k) bf.push_back((double)kfs[k]->mbf);   // e->bf = pKF->mbf, :152

    slamit_ba_problem prob;
    prob.n_kf = (int32_t)kfs.size(); prob.n_pt = (int32_t)mps.size(); prob.n_edge = (int32_t)ekf.size();
    prob.kf_pose = pose.data(); prob.kf_fixed = fixed.data(); prob.kf_intr = intr.data(); prob.pt_xyz = pts.data();
    prob.edge_kf = ekf.data(); prob.edge_pt = ept.data(); prob.edge_uv = uv.data(); prob.edge_inv_sigma2 = invSigma2.data();
    prob.edge_ur = hasStereo ? ur.data() : 0; prob.kf_bf = hasStereo ? bf.data() : 0;
    std::vector<double> outPose(pose.size()), outPts(pts.size() + 3), chi2(ekf.size() + 1);
    std::vector<uint8_t> outlier(ekf.size() + 1), outlier1(ekf.size() + 1);
    slamit_ba_result res;
    res.kf_pose = outPose.data(); res.pt_xyz = outPts.data(); res.edge_chi2 = chi2.data();
    res.edge_outlier = outlier.data(); res.edge_stage1_outlier = outlier1.data(); res.stats = 0;
    slamit_ba_opts o;
    o.its_robust = nIterations;                                        // optimizer.optimize(nIterations), :188
    o.its_final = 0;
    o.huber_delta = bRobust ? (double)(float)sqrt(5.99) : HUGE_VAL;    // thHuber2D, :87 ; no kernel = a delta nothing exceeds
    o.chi2_gate = 5.991;                                               // unused: there is no second stage
    o.huber_delta_stereo = bRobust ? (double)(float)sqrt(7.815) : HUGE_VAL;   // thHuber3D, :88
    o.chi2_gate_stereo = 7.815;
    o.stop = reinterpret_cast<const volatile uint8_t*>(pbStopFlag);
    if (SolvePOD(prob, o, res) != SLAMIT_OK) return;

    // ---- recover optimized data (:193-237) ----
    for (size_t k = 0; k < kfs.size(); ++k) {
        cv::Mat T(4, 4, CV_32F);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) T.template at<float>(r, c) = (float)outPose[12 * k + 3 * r + c];
            T.template at<float>(r, 3) = (float)outPose[12 * k + 9 + r];
            T.template at<float>(3, r) = 0.f;
        }
        T.template at<float>(3, 3) = 1.f;
        if (nLoopKF == 0) kfs[k]->SetPose(T);
        else { kfs[k]->mTcwGBA = T.clone(); kfs[k]->mnBAGlobalForKF = nLoopKF; }
    }
    for (size_t p = 0; p < mps.size(); ++p) {
        if (mps[p]->isBad()) continue;
        cv::Mat X(3, 1, CV_32F);
        for (int r = 0; r < 3; ++r) X.template at<float>(r, 0) = (float)outPts[3 * p + r];
        if (nLoopKF == 0) { mps[p]->SetWorldPos(X); mps[p]->UpdateNormalAndDepth(); }
        else { mps[p]->mPosGBA = X.clone(); mps[p]->mnBAGlobalForKF = nLoopKF; }
    }
}

// Access to the caller's similarity type: g2o::Sim3 by default; other types overload these two.
template <class Sim3T>
inline void slamit_shim_sim3_get(const Sim3T& S, double R[9], double t[3], double& s) {
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = S.rotation().toRotationMatrix()(r, c); t[r] = S.translation()[r]; }
    s = S.scale();
}
template <class Sim3T>
inline void slamit_shim_sim3_set(Sim3T& S, const double R[9], const double t[3], double s) {
    typename Sim3T::RotationMatrix Rm;   // (for g2o::Sim3 define this alias next to the class, or overload this function)
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rm(r, c) = R[3 * r + c];
    S = Sim3T(Rm, typename Sim3T::Translation(t[0], t[1], t[2]), s);
}

template <class KeyFrameT, class MapPointT, class Sim3T>
int Optimizer::OptimizeSim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches1, Sim3T& g2oS12, const float th2, const bool bFixScale) {
    const cv::Mat& K1 = pKF1->mK;
    const cv::Mat& K2 = pKF2->mK;
    const cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    const int N = (int)vpMatches1.size();
    const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches();
    std::vector<double> p1, p2, o1, o2, w1, w2;
    std::vector<int> vnIndexEdge;
    for (int i = 0; i < N; i++) {   // Optimizer.cc:1099-1178
        if (!vpMatches1[i]) continue;
        MapPointT* pMP1 = vpMapPoints1[i];
        MapPointT* pMP2 = vpMatches1[i];
        const int i2 = pMP2->GetIndexInKeyFrame(pKF2);
        if (!(pMP1 && pMP2)) continue;
        if (!(!pMP1->isBad() && !pMP2->isBad() && i2 >= 0)) continue;
        const cv::Mat P3D1w = pMP1->GetWorldPos(), P3D2w = pMP2->GetWorldPos();
        for (int r = 0; r < 3; ++r) {   // P3Dc = Rw * P3Dw + tw as cv::gemm computes a CV_32F product (sums in double), widened like toVector3d
            double a = 0, b = 0;
            for (int c = 0; c < 3; ++c) {
                a += (double)R1w.template at<float>(r, c) * (double)P3D1w.template at<float>(c, 0);
                b += (double)R2w.template at<float>(r, c) * (double)P3D2w.template at<float>(c, 0);
            }
            p1.push_back((double)(float)(a + (double)t1w.template at<float>(r, 0)));
            p2.push_back((double)(float)(b + (double)t2w.template at<float>(r, 0)));
        }
        const cv::KeyPoint& kpUn1 = pKF1->mvKeysUn[i];
        const cv::KeyPoint& kpUn2 = pKF2->mvKeysUn[i2];
        o1.push_back(kpUn1.pt.x); o1.push_back(kpUn1.pt.y);
        o2.push_back(kpUn2.pt.x); o2.push_back(kpUn2.pt.y);
        w1.push_back(pKF1->mvInvLevelSigma2[kpUn1.octave]);
        w2.push_back(pKF2->mvInvLevelSigma2[kpUn2.octave]);
        vnIndexEdge.push_back(i);
    }
    slamit_sim3_problem P;
    P.n = (int32_t)vnIndexEdge.size();
    P.p1 = p1.data(); P.p2 = p2.data(); P.obs1 = o1.data(); P.obs2 = o2.data(); P.inv_sigma2_1 = w1.data(); P.inv_sigma2_2 = w2.data();
    P.intr1[0] = K1.template at<float>(0, 0); P.intr1[1] = K1.template at<float>(1, 1); P.intr1[2] = K1.template at<float>(0, 2); P.intr1[3] = K1.template at<float>(1, 2);
    P.intr2[0] = K2.template at<float>(0, 0); P.intr2[1] = K2.template at<float>(1, 1); P.intr2[2] = K2.template at<float>(0, 2); P.intr2[3] = K2.template at<float>(1, 2);
    slamit_shim_sim3_get(g2oS12, P.r12, P.t12, P.s12);
    P.th2 = th2; P.fix_scale = bFixScale ? 1 : 0;
    std::vector<uint8_t> inlier(vnIndexEdge.size() + 1);
    slamit_sim3_result R;
    R.inlier = inlier.data();
    if ((lastStatus() = slamit_sim3_optimize(deviceRef(), &P, &R)) != SLAMIT_OK) return 0;
    for (size_t k = 0; k < vnIndexEdge.size(); ++k)
        if (!inlier[k]) vpMatches1[vnIndexEdge[k]] = static_cast<MapPointT*>(NULL);
    if (R.n_its[1] > 0 || R.n_inliers > 0) slamit_shim_sim3_set(g2oS12, R.r12, R.t12, R.s12);   // the early return 0 leaves g2oS12 alone (:1212-1213)
    return R.n_inliers;
}

template <class FrameT>
int Optimizer::PoseOptimization(FrameT* pFrame) {
    const int N = pFrame->N;
    std::vector<double> xw, uv, isg, ur;
    std::vector<int> index;
    bool hasStereo = false;
    for (int i = 0; i < N; ++i) {
        if (!pFrame->mvpMapPoints[i]) continue;
        const float kp_ur = pFrame->mvuRight[i];            // < 0: monocular observation (:281), else the stereo edge of :319-356
        ur.push_back(kp_ur < 0 ? -1.0 : (double)kp_ur);
        hasStereo = hasStereo || !(kp_ur < 0);
        pFrame->mvbOutlier[i] = false;
        const cv::KeyPoint& kpUn = pFrame->mvKeysUn[i];
        cv::Mat Xw = pFrame->mvpMapPoints[i]->GetWorldPos();
        for (int r = 0; r < 3; ++r) xw.push_back((double)Xw.template at<float>(r, 0));
        uv.push_back(kpUn.pt.x); uv.push_back(kpUn.pt.y);
        isg.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
        index.push_back(i);
    }
    if ((int)index.size() < 3) return 0;  // Optimizer.cc:364-365
    double pose[12], intr[4] = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy}, out[12];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) pose[3 * r + c] = (double)pFrame->mTcw.template at<float>(r, c);
        pose[9 + r] = (double)pFrame->mTcw.template at<float>(r, 3);
    }
    std::vector<uint8_t> outlier(index.size());
    slamit_pose_problem P;
    P.n = (int32_t)index.size(); P.pose = pose; P.intr = intr; P.xw = xw.data(); P.uv = uv.data(); P.inv_sigma2 = isg.data();
    P.ur = hasStereo ? ur.data() : 0;
    P.bf = hasStereo ? (double)pFrame->mbf : 0.0;          // e->bf = pFrame->mbf, :346 (read only when a keypoint has a right-image column)
    slamit_pose_result R;
    R.pose = out; R.outlier = outlier.data();
    if ((lastStatus() = slamit_pose_optimize(deviceRef(), &P, &R)) != SLAMIT_OK) return 0;
    for (size_t k = 0; k < index.size(); ++k) pFrame->mvbOutlier[index[k]] = outlier[k] != 0;
    cv::Mat T(4, 4, CV_32F);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T.template at<float>(r, c) = (float)out[3 * r + c];
        T.template at<float>(r, 3) = (float)out[9 + r];
        T.template at<float>(3, r) = 0.f;
    }
    T.template at<float>(3, 3) = 1.f;
    pFrame->SetPose(T);
    return R.n_inliers;
}

template <class KeyFrameT, class MapT>
void Optimizer::LocalBundleAdjustment(KeyFrameT* pKF, bool* pbStopFlag, MapT* pMap) {
    typedef decltype(pKF->GetMapPointMatches()) MapPointVec;
    typedef typename MapPointVec::value_type MapPointPtr;

    // ---- (1) local keyframes, their map points, fixed keyframes (Optimizer.cc:456-504) ----
    std::list<KeyFrameT*> lLocalKeyFrames;
    lLocalKeyFrames.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    const std::vector<KeyFrameT*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
    for (size_t i = 0; i < vNeighKFs.size(); ++i) {
        KeyFrameT* pKFi = vNeighKFs[i];
        pKFi->mnBALocalForKF = pKF->mnId;
        if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
    }
    std::list<MapPointPtr> lLocalMapPoints;
    for (typename std::list<KeyFrameT*>::iterator lit = lLocalKeyFrames.begin(); lit != lLocalKeyFrames.end(); ++lit) {
        MapPointVec vpMPs = (*lit)->GetMapPointMatches();
        for (size_t i = 0; i < vpMPs.size(); ++i) {
            MapPointPtr pMP = vpMPs[i];
            if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) {
                lLocalMapPoints.push_back(pMP);
                pMP->mnBALocalForKF = pKF->mnId;
            }
        }
    }
    std::list<KeyFrameT*> lFixedCameras;
    for (typename std::list<MapPointPtr>::iterator lit = lLocalMapPoints.begin(); lit != lLocalMapPoints.end(); ++lit) {
        std::map<KeyFrameT*, size_t> observations = (*lit)->GetObservations();
        for (typename std::map<KeyFrameT*, size_t>::iterator mit = observations.begin(); mit != observations.end(); ++mit) {
            KeyFrameT* pKFi = mit->first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
            }
        }
    }

    // ---- flatten into the C-ABI's arrays (vertices :522-546, edges :572-653) ----
    std::vector<KeyFrameT*> kfs;
    std::map<KeyFrameT*, int> kfIndex;
    std::vector<double> pose, intr, pts, uv, invSigma2, ur, bf;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> ekf, ept;
    for (int pass = 0; pass < 2; ++pass) {
        std::list<KeyFrameT*>& src = pass == 0 ? lLocalKeyFrames : lFixedCameras;
        for (typename std::list<KeyFrameT*>::iterator lit = src.begin(); lit != src.end(); ++lit) {
            KeyFrameT* k = *lit;
            kfIndex[k] = (int)kfs.size();
            kfs.push_back(k);
            cv::Mat T = k->GetPose();  // 4x4 CV_32F, widened like Converter::toSE3Quat
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) pose.push_back((double)T.template at<float>(r, c));
            for (int r = 0; r < 3; ++r) pose.push_back((double)T.template at<float>(r, 3));
            fixed.push_back(pass == 1 || k->mnId == 0);
            intr.push_back(k->fx); intr.push_back(k->fy); intr.push_back(k->cx); intr.push_back(k->cy);
        }
    }
    std::vector<MapPointPtr> mps(lLocalMapPoints.begin(), lLocalMapPoints.end());
    std::vector<std::pair<KeyFrameT*, MapPointPtr> > edgeOwner;
    bool hasStereo = false;
    for (size_t p = 0; p < mps.size(); ++p) {
        cv::Mat X = mps[p]->GetWorldPos();
        for (int r = 0; r < 3; ++r) pts.push_back((double)X.template at<float>(r, 0));
        const std::map<KeyFrameT*, size_t> observations = mps[p]->GetObservations();
        for (typename std::map<KeyFrameT*, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
            KeyFrameT* pKFi = mit->first;
            if (pKFi->isBad()) continue;
            typename std::map<KeyFrameT*, int>::iterator where = kfIndex.find(pKFi);
            if (where == kfIndex.end()) continue;
            const cv::KeyPoint& kpUn = pKFi->mvKeysUn[mit->second];
            ekf.push_back(where->second); ept.push_back((int32_t)p);
            uv.push_back(kpUn.pt.x); uv.push_back(kpUn.pt.y);
            invSigma2.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);
            const float kp_ur = pKFi->mvuRight[mit->second];         // :596 monocular below zero, :621-650 the stereo edge
            ur.push_back(kp_ur < 0 ? -1.0 : (double)kp_ur);
            hasStereo = hasStereo || !(kp_ur < 0);
            edgeOwner.push_back(std::make_pair(pKFi, mps[p]));
        }
    }
    if (hasStereo) for (size_t k2 = 0; k2 < kfs.size(); ++k2) bf.push_back((double)kfs[k2]->mbf);   // e->bf = pKFi->mbf, :641
    if (pbStopFlag && *pbStopFlag) return;  // :655-657

    // ---- (2) the optimisation itself ----
    slamit_ba_problem prob;
    prob.n_kf = (int32_t)kfs.size(); prob.n_pt = (int32_t)mps.size(); prob.n_edge = (int32_t)ekf.size();
    prob.kf_pose = pose.data(); prob.kf_fixed = fixed.data(); prob.kf_intr = intr.data(); prob.pt_xyz = pts.data();
    prob.edge_kf = ekf.data(); prob.edge_pt = ept.data(); prob.edge_uv = uv.data(); prob.edge_inv_sigma2 = invSigma2.data();
    prob.edge_ur = hasStereo ? ur.data() : 0; prob.kf_bf = hasStereo ? bf.data() : 0;
    std::vector<double> outPose(pose.size()), outPts(pts.size()), chi2(ekf.size());
    std::vector<uint8_t> outlier(ekf.size()), outlier1(ekf.size());
    slamit_ba_result res;
    res.kf_pose = outPose.data(); res.pt_xyz = outPts.data(); res.edge_chi2 = chi2.data();
    res.edge_outlier = outlier.data(); res.edge_stage1_outlier = outlier1.data(); res.stats = 0;
    if (LocalBundleAdjustmentPOD(prob, pbStopFlag, res) != SLAMIT_OK) return;

    // ---- (3) erase outlier observations, write back (:711-777) ----
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
    for (size_t e = 0; e < edgeOwner.size(); ++e) {
        if (!outlier[e] || edgeOwner[e].second->isBad()) continue;
        edgeOwner[e].first->EraseMapPointMatch(edgeOwner[e].second);
        edgeOwner[e].second->EraseObservation(edgeOwner[e].first);
    }
    size_t k = 0;
    for (typename std::list<KeyFrameT*>::iterator lit = lLocalKeyFrames.begin(); lit != lLocalKeyFrames.end(); ++lit, ++k) {
        cv::Mat T(4, 4, CV_32F);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) T.template at<float>(r, c) = (float)outPose[12 * k + 3 * r + c];
            T.template at<float>(r, 3) = (float)outPose[12 * k + 9 + r];
            T.template at<float>(3, r) = 0.f;
        }
        T.template at<float>(3, 3) = 1.f;
        (*lit)->SetPose(T);
    }
    for (size_t p = 0; p < mps.size(); ++p) {
        cv::Mat X(3, 1, CV_32F);
        for (int r = 0; r < 3; ++r) X.template at<float>(r, 0) = (float)outPts[3 * p + r];
        mps[p]->SetWorldPos(X);
        mps[p]->UpdateNormalAndDepth();
    }
}

}  // namespace ORB_SLAM2

#endif
