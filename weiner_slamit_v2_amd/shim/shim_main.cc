// shim_main.cc — exercises the three class surfaces the way the reference's callers do, with
// mock KeyFrame / MapPoint / Map types standing in for the reference's data model (which is out
// of scope and is not copied).  Driven by tests/test_shim.py:
//     shim_test orb   <in.raw> <w> <h> <out.bin>
//     shim_test match <descA.bin> <nA> <descB.bin> <nB> <out.bin>
//     shim_test ba    <problem.bin> <out.bin>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "Optimizer.h"

using namespace ORB_SLAM2;

static std::vector<unsigned char> slurp(const char* path) {
    std::vector<unsigned char> v;
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n);
    if (n && fread(v.data(), 1, n, f) != (size_t)n) exit(2);
    fclose(f);
    return v;
}

// ---- mock data model: just the members Optimizer.h documents --------------------------------
struct MockKeyFrame;
struct MockMapPoint {
    long unsigned int mnId, mnBALocalForKF;
    cv::Mat pos;
    std::map<MockKeyFrame*, size_t> obs;
    bool bad;
    int normalUpdates;
    MockMapPoint() : mnId(0), mnBALocalForKF(~0ul), bad(false), normalUpdates(0) {}
    bool isBad() { return bad; }
    std::map<MockKeyFrame*, size_t> GetObservations() { return obs; }
    cv::Mat GetWorldPos() { return pos.clone(); }
    void SetWorldPos(const cv::Mat& p) { pos = p.clone(); }
    void UpdateNormalAndDepth() { ++normalUpdates; }
    void EraseObservation(MockKeyFrame* kf) { obs.erase(kf); }
};
struct MockKeyFrame {
    long unsigned int mnId, mnBALocalForKF, mnBAFixedForKF;
    cv::Mat Tcw;
    float fx, fy, cx, cy;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight, mvInvLevelSigma2;
    std::vector<MockMapPoint*> matches;
    std::vector<MockKeyFrame*> covisible;
    int erased;
    MockKeyFrame() : mnId(0), mnBALocalForKF(~0ul), mnBAFixedForKF(~0ul), erased(0) {}
    bool isBad() { return false; }
    std::vector<MockKeyFrame*> GetVectorCovisibleKeyFrames() { return covisible; }
    std::vector<MockMapPoint*> GetMapPointMatches() { return matches; }
    cv::Mat GetPose() { return Tcw.clone(); }
    void SetPose(const cv::Mat& T) { Tcw = T.clone(); }
    void EraseMapPointMatch(MockMapPoint* mp) {
        for (size_t i = 0; i < matches.size(); ++i) if (matches[i] == mp) { matches[i] = 0; ++erased; }
    }
};
struct MockMap { std::mutex mMutexMapUpdate; };

static int run_orb(int argc, char** argv) {
    if (argc < 6) return 2;
    const int w = atoi(argv[3]), h = atoi(argv[4]);
    std::vector<unsigned char> raw = slurp(argv[2]);
    cv::Mat im(h, w, CV_8UC1, raw.data());
    ORBextractor ext(1000, 1.2f, 8, 20, 7);     // Tracking.cc:149-156
    ext.SetPyramidExport(true);
    std::vector<cv::KeyPoint> keys;
    cv::Mat desc;
    ext(im, cv::Mat(), keys, desc);
    if (!ext.ok()) { fprintf(stderr, "extract failed: %s\n", ext.lastError()); return 1; }
    FILE* f = fopen(argv[5], "wb");
    int n = (int)keys.size();
    fwrite(&n, 4, 1, f);
    if (n) fwrite(&keys[0], sizeof(cv::KeyPoint), n, f);
    for (int i = 0; i < n; ++i) fwrite(desc.ptr(i), 1, 32, f);
    // the mvImagePyramid member: ROI inside a REFLECT_101-padded plane
    int nl = ext.GetLevels();
    fwrite(&nl, 4, 1, f);
    for (int l = 0; l < nl; ++l) {
        const cv::Mat& m = ext.mvImagePyramid[l];
        int dims[2] = {m.cols, m.rows};
        fwrite(dims, 4, 2, f);
        unsigned sum = 0;  // checksum over ROI plus its 19-px frame, read through the ROI pointer
        for (int y = -19; y < m.rows + 19; ++y)
            for (int x = -19; x < m.cols + 19; ++x) sum = sum * 31u + m.data[(long)y * (long)m.step + x];
        fwrite(&sum, 4, 1, f);
    }
    std::vector<float> sf = ext.GetScaleFactors(), isig = ext.GetInverseScaleSigmaSquares();
    fwrite(sf.data(), 4, nl, f);
    fwrite(isig.data(), 4, nl, f);
    fclose(f);
    return 0;
}

static int run_match(int argc, char** argv) {
    if (argc < 7) return 2;
    std::vector<unsigned char> a = slurp(argv[2]), b = slurp(argv[4]);
    const int na = atoi(argv[3]), nb = atoi(argv[5]);
    cv::Mat da(na, 32, CV_8U, a.data()), db(nb, 32, CV_8U, b.data());
    std::vector<int> idx, best, second;
    if (!ORBmatcher::BestTwo(da, db, idx, best, second)) return 1;
    ORBmatcher m(0.9f, false);
    std::vector<cv::KeyPoint> none;
    std::vector<int> m12;
    int nm = m.SearchBruteForce(none, da, none, db, m12);
    int d00 = ORBmatcher::DescriptorDistance(da.row(0), db.row(0));
    FILE* f = fopen(argv[6], "wb");
    fwrite(idx.data(), 4, na, f); fwrite(best.data(), 4, na, f); fwrite(second.data(), 4, na, f);
    fwrite(m12.data(), 4, na, f); fwrite(&nm, 4, 1, f); fwrite(&d00, 4, 1, f);
    fclose(f);
    return 0;
}

// problem.bin: int32 n_kf n_pt n_edge | float pose[n_kf*12] | u8 fixed[n_kf] (padded to 4) |
//              float intr[4] | float pts[n_pt*3] | int32 ekf[] | int32 ept[] | float uv[2e] | float invsig[e]
static int run_ba(int argc, char** argv) {
    if (argc < 4) return 2;
    std::vector<unsigned char> raw = slurp(argv[2]);
    const unsigned char* p = raw.data();
    int hdr[3];
    memcpy(hdr, p, 12); p += 12;
    const int K = hdr[0], P = hdr[1], E = hdr[2];
    const float* pose = (const float*)p; p += 4 * 12 * K;
    const unsigned char* fixed = p; p += (K + 3) / 4 * 4;
    const float* intr = (const float*)p; p += 16;
    const float* pts = (const float*)p; p += 4 * 3 * P;
    const int* ekf = (const int*)p; p += 4 * E;
    const int* ept = (const int*)p; p += 4 * E;
    const float* uv = (const float*)p; p += 8 * E;
    const float* isg = (const float*)p;
    std::vector<MockKeyFrame> kfs(K);
    std::vector<MockMapPoint> mps(P);
    for (int k = 0; k < K; ++k) {
        MockKeyFrame& kf = kfs[k];
        // keyframe 0 of the file plays mnId 0 (fixed by id); further fixed ones are non-covisible observers
        kf.mnId = k;
        kf.Tcw = cv::Mat(4, 4, CV_32F);
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) kf.Tcw.at<float>(r, c) = pose[12 * k + 3 * r + c]; kf.Tcw.at<float>(r, 3) = pose[12 * k + 9 + r]; kf.Tcw.at<float>(3, r) = 0; }
        kf.Tcw.at<float>(3, 3) = 1;
        kf.fx = intr[0]; kf.fy = intr[1]; kf.cx = intr[2]; kf.cy = intr[3];
        kf.mvInvLevelSigma2.assign(1, 1.0f);
    }
    for (int q = 0; q < P; ++q) {
        mps[q].mnId = q;
        mps[q].pos = cv::Mat(3, 1, CV_32F);
        for (int r = 0; r < 3; ++r) mps[q].pos.at<float>(r, 0) = pts[3 * q + r];
    }
    for (int e = 0; e < E; ++e) {
        MockKeyFrame& kf = kfs[ekf[e]];
        size_t slot = kf.mvKeysUn.size();
        cv::KeyPoint kp(uv[2 * e], uv[2 * e + 1], 31.f);
        kp.octave = (int)kf.mvInvLevelSigma2.size();   // one sigma entry per observation keeps invSigma2 exact
        kf.mvInvLevelSigma2.push_back(isg[e]);
        kf.mvKeysUn.push_back(kp);
        kf.mvuRight.push_back(-1.f);
        kf.matches.push_back(&mps[ept[e]]);
        mps[ept[e]].obs[&kf] = slot;
    }
    // the current keyframe is the last non-fixed one; every other non-fixed keyframe is covisible
    int cur = -1;
    for (int k = 0; k < K; ++k) if (!fixed[k] || k == 0) cur = k;
    for (int k = 0; k < K; ++k) if (k != cur && (!fixed[k] || k == 0)) kfs[cur].covisible.push_back(&kfs[k]);
    MockMap map;
    bool stop = false;
    Optimizer::LocalBundleAdjustment(&kfs[cur], &stop, &map);
    if (Optimizer::LastStatus() != 0) { fprintf(stderr, "BA failed: %s\n", slamit_last_error()); return 1; }
    FILE* f = fopen(argv[3], "wb");
    for (int k = 0; k < K; ++k)
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) fwrite(&kfs[k].Tcw.at<float>(r, c), 4, 1, f); }
    for (int k = 0; k < K; ++k) for (int r = 0; r < 3; ++r) fwrite(&kfs[k].Tcw.at<float>(r, 3), 4, 1, f);
    for (int q = 0; q < P; ++q) for (int r = 0; r < 3; ++r) fwrite(&mps[q].pos.at<float>(r, 0), 4, 1, f);
    int erased = 0, updates = 0;
    for (int k = 0; k < K; ++k) erased += kfs[k].erased;
    for (int q = 0; q < P; ++q) updates += mps[q].normalUpdates;
    fwrite(&erased, 4, 1, f); fwrite(&updates, 4, 1, f);
    fclose(f);
    return 0;
}

// pose.bin: int32 n | float pose[12] | float intr[4] | float xw[3n] | float uv[2n] | float invsig[n]
struct MockFrame {
    int N;
    cv::Mat mTcw;
    float fx, fy, cx, cy;
    std::vector<MockMapPoint*> mvpMapPoints;
    std::vector<float> mvuRight, mvInvLevelSigma2;
    std::vector<bool> mvbOutlier;
    std::vector<cv::KeyPoint> mvKeysUn;
    void SetPose(const cv::Mat& T) { mTcw = T.clone(); }
};

static int run_pose(int argc, char** argv) {
    if (argc < 4) return 2;
    std::vector<unsigned char> raw = slurp(argv[2]);
    const unsigned char* p = raw.data();
    int n;
    memcpy(&n, p, 4); p += 4;
    const float* pose = (const float*)p; p += 48;
    const float* intr = (const float*)p; p += 16;
    const float* xw = (const float*)p; p += 12 * n;
    const float* uv = (const float*)p; p += 8 * n;
    const float* isg = (const float*)p;
    MockFrame F;
    std::vector<MockMapPoint> mps(n);
    F.N = n + 5;  // a few keypoints without a map point, like a real frame
    F.mTcw = cv::Mat(4, 4, CV_32F);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) F.mTcw.at<float>(r, c) = pose[3 * r + c]; F.mTcw.at<float>(r, 3) = pose[9 + r]; F.mTcw.at<float>(3, r) = 0; }
    F.mTcw.at<float>(3, 3) = 1;
    F.fx = intr[0]; F.fy = intr[1]; F.cx = intr[2]; F.cy = intr[3];
    F.mvpMapPoints.assign(F.N, (MockMapPoint*)0); F.mvuRight.assign(F.N, -1.f); F.mvbOutlier.assign(F.N, false);
    F.mvKeysUn.resize(F.N); F.mvInvLevelSigma2.assign(1, 1.f);
    for (int i = 0; i < n; ++i) {
        mps[i].pos = cv::Mat(3, 1, CV_32F);
        for (int r = 0; r < 3; ++r) mps[i].pos.at<float>(r, 0) = xw[3 * i + r];
        F.mvpMapPoints[i] = &mps[i];
        F.mvKeysUn[i] = cv::KeyPoint(uv[2 * i], uv[2 * i + 1], 31.f);
        F.mvKeysUn[i].octave = (int)F.mvInvLevelSigma2.size();
        F.mvInvLevelSigma2.push_back(isg[i]);
    }
    int inl = Optimizer::PoseOptimization(&F);
    if (Optimizer::LastStatus() != 0) { fprintf(stderr, "pose failed: %s\n", slamit_last_error()); return 1; }
    FILE* f = fopen(argv[3], "wb");
    fwrite(&inl, 4, 1, f);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) fwrite(&F.mTcw.at<float>(r, c), 4, 1, f);
    for (int i = 0; i < n; ++i) { unsigned char o = F.mvbOutlier[i]; fwrite(&o, 1, 1, f); }
    fclose(f);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::string mode = argv[1];
    if (mode == "orb") return run_orb(argc, argv);
    if (mode == "match") return run_match(argc, argv);
    if (mode == "ba") return run_ba(argc, argv);
    if (mode == "pose") return run_pose(argc, argv);
    return 2;
}
