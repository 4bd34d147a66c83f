// shim_main.cc — exercises the three class surfaces the way the reference's callers do, with
// mock KeyFrame / MapPoint / Map types standing in for the reference's data model (which is out
// of scope and is not copied).  Driven by tests/test_shim.py:
//     shim_test orb   <in.raw> <w> <h> <out.bin>
//     shim_test match <descA.bin> <nA> <descB.bin> <nB> <out.bin>
//     shim_test ba    <problem.bin> <out.bin>
//     shim_test pose  <problem.bin> <out.bin>
//     shim_test search <problem.bin> <out.bin>
//     shim_test frame <problem.bin> <out.bin>
//     shim_test fuse <problem.bin> <out.bin>
//     shim_test init <problem.bin> <out.bin>
//     shim_test bow <problem.bin> <out.bin>
//     shim_test sim3 <problem.bin> <out.bin>
//     shim_test osim3 <problem.bin> <out.bin>
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
#include "FrameOps.h"

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
    cv::Mat mPosGBA;
    long unsigned int mnBAGlobalForKF;
    MockMapPoint() : mnId(0), mnBALocalForKF(~0ul), bad(false), normalUpdates(0), mnBAGlobalForKF(0) {}
    bool isBad() { return bad; }
    std::map<MockKeyFrame*, size_t> GetObservations() { return obs; }
    cv::Mat GetWorldPos() { return pos.clone(); }
    void SetWorldPos(const cv::Mat& p) { pos = p.clone(); }
    void UpdateNormalAndDepth() { ++normalUpdates; }
    void EraseObservation(MockKeyFrame* kf) { obs.erase(kf); }
};
struct MockKeyFrame {
    long unsigned int mnId, mnBALocalForKF, mnBAFixedForKF, mnBAGlobalForKF;
    cv::Mat Tcw, mTcwGBA;
    float fx, fy, cx, cy, mbf;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight, mvInvLevelSigma2;
    std::vector<MockMapPoint*> matches;
    std::vector<MockKeyFrame*> covisible;
    int erased;
    MockKeyFrame() : mnId(0), mnBALocalForKF(~0ul), mnBAFixedForKF(~0ul), mnBAGlobalForKF(0), mbf(0.f), erased(0) {}
    bool isBad() { return false; }
    std::vector<MockKeyFrame*> GetVectorCovisibleKeyFrames() { return covisible; }
    std::vector<MockMapPoint*> GetMapPointMatches() { return matches; }
    cv::Mat GetPose() { return Tcw.clone(); }
    void SetPose(const cv::Mat& T) { Tcw = T.clone(); }
    void EraseMapPointMatch(MockMapPoint* mp) {
        for (size_t i = 0; i < matches.size(); ++i) if (matches[i] == mp) { matches[i] = 0; ++erased; }
    }
};
struct MockMap {
    std::mutex mMutexMapUpdate;
    std::vector<MockKeyFrame*> kfs;
    std::vector<MockMapPoint*> mps;
    std::vector<MockKeyFrame*> GetAllKeyFrames() { return kfs; }
    std::vector<MockMapPoint*> GetAllMapPoints() { return mps; }
};

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
//              [ | float bf | float ur[e] ]   (a window with stereo observations: KeyFrame::mbf and mvuRight, -1 = monocular)
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
    const float* isg = (const float*)p; p += 4 * E;
    const bool stereo = (size_t)(p - raw.data()) + 4 + 4 * (size_t)E <= raw.size();
    const float bf = stereo ? *(const float*)p : 0.f;
    const float* ur = stereo ? (const float*)(p + 4) : 0;
    std::vector<MockKeyFrame> kfs(K);
    std::vector<MockMapPoint> mps(P);
    for (int k = 0; k < K; ++k) {
        MockKeyFrame& kf = kfs[k];
        // keyframe 0 of the file plays mnId 0 (fixed by id); further fixed ones are non-covisible observers
        kf.mnId = k;
        kf.Tcw = cv::Mat(4, 4, CV_32F);
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) kf.Tcw.at<float>(r, c) = pose[12 * k + 3 * r + c]; kf.Tcw.at<float>(r, 3) = pose[12 * k + 9 + r]; kf.Tcw.at<float>(3, r) = 0; }
        kf.Tcw.at<float>(3, 3) = 1;
        kf.fx = intr[0]; kf.fy = intr[1]; kf.cx = intr[2]; kf.cy = intr[3]; kf.mbf = bf;
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
        kf.mvuRight.push_back(ur ? ur[e] : -1.f);
        kf.matches.push_back(&mps[ept[e]]);
        mps[ept[e]].obs[&kf] = slot;
    }
    // the current keyframe is the last non-fixed one; every other non-fixed keyframe is covisible
    int cur = -1;
    for (int k = 0; k < K; ++k) if (!fixed[k] || k == 0) cur = k;
    for (int k = 0; k < K; ++k) if (k != cur && (!fixed[k] || k == 0)) kfs[cur].covisible.push_back(&kfs[k]);
    MockMap map;
    bool stop = false;
    // shim_test ba <problem> <out> [global <nIterations> <nLoopKF> <bRobust>]: GlobalBundleAdjustemnt over the same data
    const bool global = argc >= 8 && std::string(argv[4]) == "global";
    const unsigned long nLoopKF = global ? (unsigned long)atoi(argv[6]) : 0;
    if (global) {
        for (int k = 0; k < K; ++k) map.kfs.push_back(&kfs[k]);
        for (int q = 0; q < P; ++q) map.mps.push_back(&mps[q]);
        Optimizer::GlobalBundleAdjustemnt(&map, atoi(argv[5]), &stop, nLoopKF, atoi(argv[7]) != 0);
        if (nLoopKF) {   // results went to mTcwGBA / mPosGBA: fold them back so that the output format stays the same
            for (int k = 0; k < K; ++k) if (kfs[k].mnBAGlobalForKF == nLoopKF) kfs[k].Tcw = kfs[k].mTcwGBA.clone();
            for (int q = 0; q < P; ++q) if (mps[q].mnBAGlobalForKF == nLoopKF) { mps[q].pos = mps[q].mPosGBA.clone(); mps[q].normalUpdates = -1; }
        }
    } else {
        Optimizer::LocalBundleAdjustment(&kfs[cur], &stop, &map);
    }
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

// pose.bin: int32 n | float pose[12] | float intr[4] | float xw[3n] | float uv[2n] | float invsig[n] [ | float bf | float ur[n] ]
struct MockFrame {
    int N;
    cv::Mat mTcw;
    float fx, fy, cx, cy, mbf;
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
    const float* isg = (const float*)p; p += 4 * n;
    // optional tail: float bf | float ur[n]  (a frame with stereo keypoints: Frame::mbf, mvuRight)
    const bool stereo = (size_t)(p - raw.data()) + 4 + 4 * (size_t)n <= raw.size();
    const float* ur = stereo ? (const float*)(p + 4) : 0;
    MockFrame F;
    F.mbf = stereo ? *(const float*)p : 0.f;
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
        if (ur) F.mvuRight[i] = ur[i];
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

// ---- guided search: mock Frame / MapPoint with the members ORBmatcher.h documents -------------------
struct MockTrackPoint {
    bool mbTrackInView, bad;
    int mnTrackScaleLevel, nobs, id;
    float mTrackViewCos, mTrackProjX, mTrackProjY, maxd, mind;
    cv::Mat desc, pos;
    float GetMaxDistanceInvariance() { return maxd; }
    float GetMinDistanceInvariance() { return mind; }
    int PredictScale(const float&, const float&) { return mnTrackScaleLevel; }   // the caller's method: stored level
    bool isBad() { return bad; }
    int Observations() { return nobs; }
    cv::Mat GetDescriptor() { return desc.clone(); }
    cv::Mat GetWorldPos() { return pos.clone(); }
};
struct MockSearchFrame {
    int N;
    cv::Mat mTcw, mDescriptors;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<MockTrackPoint*> mvpMapPoints;
    std::vector<float> mvuRight, mvScaleFactors;
    std::vector<bool> mvbOutlier;
    float fx, fy, cx, cy, mb, mfLogScaleFactor;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
};
struct MockRelocKF {
    std::vector<MockTrackPoint*> matches;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<MockTrackPoint*> GetMapPointMatches() { return matches; }
};
float MockSearchFrame::mnMinX, MockSearchFrame::mnMaxX, MockSearchFrame::mnMinY, MockSearchFrame::mnMaxY;
float MockSearchFrame::mfGridElementWidthInv, MockSearchFrame::mfGridElementHeightInv;

struct Reader {
    const unsigned char* p;
    template <class T> T get() { T v; memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
    template <class T> const T* arr(size_t n) { const T* a = (const T*)p; p += n * sizeof(T); return a; }
};

static cv::Mat mat_from(const float* v, int rows) {
    cv::Mat m(rows, 1, CV_32F);
    for (int r = 0; r < rows; ++r) m.at<float>(r, 0) = v[r];
    return m;
}
static cv::Mat pose_from(const float* p12) {   // R row-major (9) then t (3)
    cv::Mat T = cv::Mat::zeros(4, 4, CV_32F);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T.at<float>(r, c) = p12[3 * r + c]; T.at<float>(r, 3) = p12[9 + r]; }
    T.at<float>(3, 3) = 1;
    return T;
}

// problem.bin (all little-endian, 4-byte aligned arrays first):
//   int32 variant n m ; float th nnratio ; float minx maxx miny maxy invw invh ; float scale[8]
//   current frame: float xy[2n] ; int32 octave[n] ; float angle[n] ; int32 state[n] (0 none, 1 map point with
//   observations, 2 map point without) ; u8 desc[32n]
//   variant 0 (map points): float proj[2m] viewcos[m] ; int32 level[m] inview[m] bad[m] nobs[m] ; u8 desc[32m]
//   variant 1 (last frame): float Tcw[12] Tlw[12] intr[5] ; int32 mono ; float world[3m] angle[m] ;
//                           int32 has[m] outlier[m] octave[m] nobs[m] ; u8 desc[32m]
//   variant 2 (relocalization, keyframe map points): float Tcw[12] intr[4] ; int32 orbdist ; float world[3m] angle[m] maxd[m]
//                           mind[m] ; int32 has[m] bad[m] found[m] level[m] ; u8 desc[32m]
// out.bin: int32 status nmatches ; int32 owner[n] (-1 none, -2 the frame's own earlier point, else query index)
static int run_search(int argc, char** argv) {
    if (argc < 4) return 2;
    std::vector<unsigned char> raw = slurp(argv[2]);
    Reader R{raw.data()};
    const int variant = R.get<int>(), n = R.get<int>(), m = R.get<int>();
    const float th = R.get<float>(), nnratio = R.get<float>();
    MockSearchFrame::mnMinX = R.get<float>(); MockSearchFrame::mnMaxX = R.get<float>();
    MockSearchFrame::mnMinY = R.get<float>(); MockSearchFrame::mnMaxY = R.get<float>();
    MockSearchFrame::mfGridElementWidthInv = R.get<float>(); MockSearchFrame::mfGridElementHeightInv = R.get<float>();
    const float* scale = R.arr<float>(8);
    const float* xy = R.arr<float>(2 * (size_t)n);
    const int* oct = R.arr<int>(n);
    const float* ang = R.arr<float>(n);
    const int* state = R.arr<int>(n);
    const unsigned char* desc = R.arr<unsigned char>(32 * (size_t)n);
    MockSearchFrame F;
    F.N = n;
    F.mvScaleFactors.assign(scale, scale + 8);
    F.mDescriptors = cv::Mat(n, 32, CV_8U);
    F.mvKeysUn.resize(n); F.mvuRight.assign(n, -1.f); F.mvpMapPoints.assign(n, (MockTrackPoint*)0); F.mvbOutlier.assign(n, false);
    MockTrackPoint own1, own2;
    own1.nobs = 3; own2.nobs = 0; own1.id = own2.id = -2;
    for (int i = 0; i < n; ++i) {
        F.mvKeysUn[i] = cv::KeyPoint(xy[2 * i], xy[2 * i + 1], 31.f, ang[i], 0, oct[i]);
        memcpy(F.mDescriptors.ptr(i), desc + 32 * (size_t)i, 32);
        if (state[i] == 1) F.mvpMapPoints[i] = &own1;
        if (state[i] == 2) F.mvpMapPoints[i] = &own2;
    }
    F.mvKeys = F.mvKeysUn;
    std::vector<MockTrackPoint> mps(m);
    int nm = 0;
    if (variant == 0) {
        const float* proj = R.arr<float>(2 * (size_t)m);
        const float* vc = R.arr<float>(m);
        const int* level = R.arr<int>(m); const int* inview = R.arr<int>(m); const int* bad = R.arr<int>(m); const int* nobs = R.arr<int>(m);
        const unsigned char* qd = R.arr<unsigned char>(32 * (size_t)m);
        std::vector<MockTrackPoint*> vp(m);
        for (int q = 0; q < m; ++q) {
            MockTrackPoint& p = mps[q];
            p.id = q; p.mbTrackInView = inview[q] != 0; p.bad = bad[q] != 0; p.mnTrackScaleLevel = level[q]; p.nobs = nobs[q];
            p.mTrackViewCos = vc[q]; p.mTrackProjX = proj[2 * q]; p.mTrackProjY = proj[2 * q + 1];
            p.desc = cv::Mat(1, 32, CV_8U); memcpy(p.desc.ptr(0), qd + 32 * (size_t)q, 32);
            vp[q] = &p;
        }
        ORBmatcher matcher(nnratio, true);
        nm = matcher.SearchByProjection(F, vp, th);
    } else if (variant == 2) {
        const float* Tcw = R.arr<float>(12); const float* intr = R.arr<float>(4);
        const int orbdist = R.get<int>();
        const float* world = R.arr<float>(3 * (size_t)m); const float* kang = R.arr<float>(m);
        const float* maxd = R.arr<float>(m); const float* mind = R.arr<float>(m);
        const int* has = R.arr<int>(m); const int* bad = R.arr<int>(m); const int* found = R.arr<int>(m); const int* level = R.arr<int>(m);
        const unsigned char* qd = R.arr<unsigned char>(32 * (size_t)m);
        F.mTcw = pose_from(Tcw);
        F.fx = intr[0]; F.fy = intr[1]; F.cx = intr[2]; F.cy = intr[3]; F.mfLogScaleFactor = 0.18232f;
        MockRelocKF KF;
        KF.matches.assign(m, (MockTrackPoint*)0); KF.mvKeysUn.resize(m);
        std::set<MockTrackPoint*> sFound;
        for (int q = 0; q < m; ++q) {
            MockTrackPoint& p = mps[q];
            p.id = q; p.bad = bad[q] != 0; p.mnTrackScaleLevel = level[q]; p.maxd = maxd[q]; p.mind = mind[q]; p.nobs = 1;
            p.pos = mat_from(world + 3 * (size_t)q, 3);
            p.desc = cv::Mat(1, 32, CV_8U); memcpy(p.desc.ptr(0), qd + 32 * (size_t)q, 32);
            KF.mvKeysUn[q] = cv::KeyPoint(0.f, 0.f, 31.f, kang[q], 0, 0);
            if (has[q]) KF.matches[q] = &p;
            if (found[q]) sFound.insert(&p);
        }
        ORBmatcher matcher(0.9f, true);
        nm = matcher.SearchByProjection(F, &KF, sFound, th, orbdist);
    } else {
        const float* Tcw = R.arr<float>(12); const float* Tlw = R.arr<float>(12); const float* intr = R.arr<float>(5);
        const int mono = R.get<int>();
        const float* world = R.arr<float>(3 * (size_t)m); const float* lang = R.arr<float>(m);
        const int* has = R.arr<int>(m); const int* outl = R.arr<int>(m); const int* loct = R.arr<int>(m); const int* nobs = R.arr<int>(m);
        const unsigned char* qd = R.arr<unsigned char>(32 * (size_t)m);
        MockSearchFrame L;
        L.N = m; L.mTcw = pose_from(Tlw); F.mTcw = pose_from(Tcw);
        F.fx = intr[0]; F.fy = intr[1]; F.cx = intr[2]; F.cy = intr[3]; F.mb = intr[4];
        L.mvKeys.resize(m); L.mvpMapPoints.assign(m, (MockTrackPoint*)0); L.mvbOutlier.assign(m, false);
        for (int q = 0; q < m; ++q) {
            MockTrackPoint& p = mps[q];
            p.id = q; p.nobs = nobs[q]; p.pos = mat_from(world + 3 * (size_t)q, 3);
            p.desc = cv::Mat(1, 32, CV_8U); memcpy(p.desc.ptr(0), qd + 32 * (size_t)q, 32);
            L.mvKeys[q] = cv::KeyPoint(0.f, 0.f, 31.f, lang[q], 0, loct[q]);
            if (has[q]) L.mvpMapPoints[q] = &p;
            L.mvbOutlier[q] = outl[q] != 0;
        }
        L.mvKeysUn = L.mvKeys;
        ORBmatcher matcher(0.9f, true);
        nm = matcher.SearchByProjection(F, L, th, mono != 0);
    }
    const int status = ORBmatcher::LastStatus();
    if (status != 0) fprintf(stderr, "search failed: %s\n", slamit_last_error());
    FILE* f = fopen(argv[3], "wb");
    fwrite(&status, 4, 1, f); fwrite(&nm, 4, 1, f);
    for (int i = 0; i < n; ++i) { int o = F.mvpMapPoints[i] ? F.mvpMapPoints[i]->id : -1; fwrite(&o, 4, 1, f); }
    fclose(f);
    return 0;
}

// ---- Frame epilogue: mock Frame with the members FrameOps.h documents -----------------------------------
struct MockEpiFrame {
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    cv::Mat mK, mDistCoef;
    std::vector<std::size_t> mGrid[SLAMIT_FRAME_GRID_COLS][SLAMIT_FRAME_GRID_ROWS];
    static float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
};
float MockEpiFrame::mnMinX, MockEpiFrame::mnMaxX, MockEpiFrame::mnMinY, MockEpiFrame::mnMaxY;
float MockEpiFrame::mfGridElementWidthInv, MockEpiFrame::mfGridElementHeightInv;

// problem.bin: int32 n cols rows ; float cam[9] ; slamit_kp kps[n]
// out.bin: int32 status ; float bounds[6] ; slamit_kp kps_un[n] ; int32 counts[64*48] ; int32 items[sum counts]
static int run_frame(int argc, char** argv) {
    if (argc < 4) return 2;
    std::vector<unsigned char> raw = slurp(argv[2]);
    Reader R{raw.data()};
    const int n = R.get<int>(), cols = R.get<int>(), rows = R.get<int>();
    const float* cam = R.arr<float>(9);
    const cv::KeyPoint* kps = R.arr<cv::KeyPoint>(n);
    MockEpiFrame F;
    F.N = n;
    F.mvKeys.assign(kps, kps + n);
    F.mK = cv::Mat::zeros(3, 3, CV_32F);
    F.mK.at<float>(0, 0) = cam[0]; F.mK.at<float>(1, 1) = cam[1]; F.mK.at<float>(0, 2) = cam[2]; F.mK.at<float>(1, 2) = cam[3];
    F.mK.at<float>(2, 2) = 1.f;
    F.mDistCoef = cv::Mat(5, 1, CV_32F);
    for (int i = 0; i < 5; ++i) F.mDistCoef.at<float>(i, 0) = cam[4 + i];
    cv::Mat im(rows, cols, CV_8U);
    FrameOps::ComputeImageBounds(F, im);
    int status = FrameOps::LastStatus();
    if (status == 0) { FrameOps::UndistortAndAssign(F); status = FrameOps::LastStatus(); }
    if (status != 0) fprintf(stderr, "frame failed: %s\n", slamit_last_error());
    FILE* f = fopen(argv[3], "wb");
    fwrite(&status, 4, 1, f);
    const float b[6] = {MockEpiFrame::mnMinX, MockEpiFrame::mnMaxX, MockEpiFrame::mnMinY, MockEpiFrame::mnMaxY,
                        MockEpiFrame::mfGridElementWidthInv, MockEpiFrame::mfGridElementHeightInv};
    fwrite(b, 4, 6, f);
    fwrite(F.mvKeysUn.data(), sizeof(cv::KeyPoint), F.mvKeysUn.size(), f);
    for (int x = 0; x < SLAMIT_FRAME_GRID_COLS; ++x)
        for (int y = 0; y < SLAMIT_FRAME_GRID_ROWS; ++y) { int c = (int)F.mGrid[x][y].size(); fwrite(&c, 4, 1, f); }
    for (int x = 0; x < SLAMIT_FRAME_GRID_COLS; ++x)
        for (int y = 0; y < SLAMIT_FRAME_GRID_ROWS; ++y)
            for (size_t j = 0; j < F.mGrid[x][y].size(); ++j) { int v = (int)F.mGrid[x][y][j]; fwrite(&v, 4, 1, f); }
    fclose(f);
    return 0;
}

// ---- Fuse: mock KeyFrame / MapPoint with the members ORBmatcher.h documents ---------------------------------
struct MockFuseKF;
struct MockFusePoint {
    int id, nobs, level;
    bool bad, inKF;
    float maxd, mind;
    cv::Mat pos, normal, desc;
    MockFusePoint* replacedBy;
    int addedAt, idxInKF2;
    MockFusePoint() : id(-1), nobs(0), level(0), bad(false), inKF(false), maxd(0), mind(0), replacedBy(0), addedAt(-1), idxInKF2(-1) {}
    int GetIndexInKeyFrame(MockFuseKF*) { return idxInKF2; }
    bool isBad() { return bad; }
    bool IsInKeyFrame(MockFuseKF*) { return inKF; }
    cv::Mat GetWorldPos() { return pos.clone(); }
    cv::Mat GetNormal() { return normal.clone(); }
    cv::Mat GetDescriptor() { return desc.clone(); }
    float GetMaxDistanceInvariance() { return maxd; }
    float GetMinDistanceInvariance() { return mind; }
    int PredictScale(const float&, const float&) { return level; }   // the caller's method: the mock returns a stored level
    int Observations() { return nobs; }
    void Replace(MockFusePoint* p) { replacedBy = p; bad = true; }
    void AddObservation(MockFuseKF*, size_t idx) { addedAt = (int)idx; inKF = true; ++nobs; }
};
struct MockFuseKF {
    cv::Mat R, t, O, mDescriptors, mK;
    float fx, fy, cx, cy, mfLogScaleFactor;
    float mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<float> mvScaleFactors, mvInvLevelSigma2, mvuRight;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<MockFusePoint*> mps;
    cv::Mat GetRotation() { return R.clone(); }
    cv::Mat GetTranslation() { return t.clone(); }
    cv::Mat GetCameraCenter() { return O.clone(); }
    bool IsInImage(const float& x, const float& y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }
    MockFusePoint* GetMapPoint(size_t idx) { return mps[idx]; }
    void AddMapPoint(MockFusePoint* p, size_t idx) { mps[idx] = p; }
    std::vector<MockFusePoint*> GetMapPointMatches() { return mps; }
    std::set<MockFusePoint*> GetMapPoints() { std::set<MockFusePoint*> r; for (size_t i = 0; i < mps.size(); ++i) if (mps[i]) r.insert(mps[i]); return r; }
};

// problem.bin: int32 n m ; float th ; float R[9] t[3] O[3] ; float intr[4] ; float bounds[6] (minx maxx miny maxy invw invh) ;
//   float scale[8] invsig[8] ; keypoints: float xy[2n], int32 octave[n], int32 kf_state[n] (0 none, k>0: own point with k-1
//   observations), u8 desc[32n] ; map points: float pos[3m] normal[3m] maxd[m] mind[m], int32 level[m] nobs[m] bad[m] inkf[m]
//   null[m], u8 desc[32m]
// out.bin: int32 status nFused ; per map point int32 addedAt, replacedBy (-1 none, -2 a keyframe-own point, else id), bad ;
//   per keypoint int32 owner (-1 none, -2 own, else map point id)
static int run_fuse(int argc, char** argv) {
    if (argc < 4) return 2;
    std::vector<unsigned char> raw = slurp(argv[2]);
    Reader Rd{raw.data()};
    const int n = Rd.get<int>(), m = Rd.get<int>();
    const float th = Rd.get<float>();
    const float* Rv = Rd.arr<float>(9); const float* tv = Rd.arr<float>(3); const float* Ov = Rd.arr<float>(3);
    const float* intr = Rd.arr<float>(4); const float* b = Rd.arr<float>(6);
    const float* scale = Rd.arr<float>(8); const float* invsig = Rd.arr<float>(8);
    const float* xy = Rd.arr<float>(2 * (size_t)n); const int* oct = Rd.arr<int>(n); const int* kfs = Rd.arr<int>(n);
    const unsigned char* kd = Rd.arr<unsigned char>(32 * (size_t)n);
    const float* pos = Rd.arr<float>(3 * (size_t)m); const float* nrm = Rd.arr<float>(3 * (size_t)m);
    const float* maxd = Rd.arr<float>(m); const float* mind = Rd.arr<float>(m);
    const int* level = Rd.arr<int>(m); const int* nobs = Rd.arr<int>(m); const int* bad = Rd.arr<int>(m); const int* inkf = Rd.arr<int>(m);
    const int* isnull = Rd.arr<int>(m);
    const unsigned char* md = Rd.arr<unsigned char>(32 * (size_t)m);
    MockFuseKF KF;
    KF.R = cv::Mat(3, 3, CV_32F); KF.t = cv::Mat(3, 1, CV_32F); KF.O = cv::Mat(3, 1, CV_32F);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) KF.R.at<float>(r, c) = Rv[3 * r + c]; KF.t.at<float>(r, 0) = tv[r]; KF.O.at<float>(r, 0) = Ov[r]; }
    KF.fx = intr[0]; KF.fy = intr[1]; KF.cx = intr[2]; KF.cy = intr[3]; KF.mfLogScaleFactor = 0.18232f;
    KF.mnMinX = b[0]; KF.mnMaxX = b[1]; KF.mnMinY = b[2]; KF.mnMaxY = b[3]; KF.mfGridElementWidthInv = b[4]; KF.mfGridElementHeightInv = b[5];
    KF.mvScaleFactors.assign(scale, scale + 8); KF.mvInvLevelSigma2.assign(invsig, invsig + 8);
    KF.mvuRight.assign(n, -1.f); KF.mvKeysUn.resize(n); KF.mps.assign(n, (MockFusePoint*)0);
    KF.mDescriptors = cv::Mat(n, 32, CV_8U);
    std::vector<MockFusePoint> own(n), pts(m);
    for (int i = 0; i < n; ++i) {
        KF.mvKeysUn[i] = cv::KeyPoint(xy[2 * i], xy[2 * i + 1], 31.f, -1.f, 0, oct[i]);
        memcpy(KF.mDescriptors.ptr(i), kd + 32 * (size_t)i, 32);
        if (kfs[i] > 0) { own[i].id = -2; own[i].nobs = kfs[i] - 1; KF.mps[i] = &own[i]; }
    }
    std::vector<MockFusePoint*> vp(m);
    for (int j = 0; j < m; ++j) {
        MockFusePoint& p = pts[j];
        p.id = j; p.nobs = nobs[j]; p.level = level[j]; p.bad = bad[j] != 0; p.inKF = inkf[j] != 0; p.maxd = maxd[j]; p.mind = mind[j];
        p.pos = mat_from(pos + 3 * (size_t)j, 3); p.normal = mat_from(nrm + 3 * (size_t)j, 3);
        p.desc = cv::Mat(1, 32, CV_8U); memcpy(p.desc.ptr(0), md + 32 * (size_t)j, 32);
        vp[j] = isnull[j] ? (MockFusePoint*)0 : &p;
    }
    ORBmatcher matcher(0.6f, true);
    const int nf = matcher.Fuse(&KF, vp, th);
    const int status = ORBmatcher::LastStatus();
    if (status != 0) fprintf(stderr, "fuse failed: %s\n", slamit_last_error());
    FILE* f = fopen(argv[3], "wb");
    fwrite(&status, 4, 1, f); fwrite(&nf, 4, 1, f);
    for (int j = 0; j < m; ++j) {
        int a = pts[j].addedAt, r = pts[j].replacedBy ? pts[j].replacedBy->id : -1, bd = pts[j].bad ? 1 : 0;
        fwrite(&a, 4, 1, f); fwrite(&r, 4, 1, f); fwrite(&bd, 4, 1, f);
    }
    for (int i = 0; i < n; ++i) { int o = KF.mps[i] ? KF.mps[i]->id : -1; fwrite(&o, 4, 1, f); }
    for (int i = 0; i < n; ++i) { int r = own[i].replacedBy ? own[i].replacedBy->id : -1; fwrite(&r, 4, 1, f); }
    fclose(f);
    return 0;
}

// problem.bin: int32 n1 n2 window ; float nnratio ; float bounds[6] ; F1: int32 octave[n1], float angle[n1], float prev[2 n1],
//   u8 desc[32 n1] ; F2: float xy[2 n2], int32 octave[n2], float angle[n2], u8 desc[32 n2]
// out.bin: int32 status nmatches ; int32 vnMatches12[n1] ; float prev[2 n1]
static int run_init(int argc, char** argv) {
    if (argc < 4) return 2;
    std::vector<unsigned char> raw = slurp(argv[2]);
    Reader R{raw.data()};
    const int n1 = R.get<int>(), n2 = R.get<int>(), window = R.get<int>();
    const float nnratio = R.get<float>();
    MockSearchFrame::mnMinX = R.get<float>(); MockSearchFrame::mnMaxX = R.get<float>();
    MockSearchFrame::mnMinY = R.get<float>(); MockSearchFrame::mnMaxY = R.get<float>();
    MockSearchFrame::mfGridElementWidthInv = R.get<float>(); MockSearchFrame::mfGridElementHeightInv = R.get<float>();
    const int* o1 = R.arr<int>(n1); const float* a1 = R.arr<float>(n1); const float* prev = R.arr<float>(2 * (size_t)n1);
    const unsigned char* d1 = R.arr<unsigned char>(32 * (size_t)n1);
    const float* xy2 = R.arr<float>(2 * (size_t)n2); const int* o2 = R.arr<int>(n2); const float* a2 = R.arr<float>(n2);
    const unsigned char* d2 = R.arr<unsigned char>(32 * (size_t)n2);
    MockSearchFrame F1, F2;
    F1.N = n1; F2.N = n2;
    F1.mDescriptors = cv::Mat(n1, 32, CV_8U); F2.mDescriptors = cv::Mat(n2, 32, CV_8U);
    F1.mvKeysUn.resize(n1); F2.mvKeysUn.resize(n2);
    for (int i = 0; i < n1; ++i) { F1.mvKeysUn[i] = cv::KeyPoint(0.f, 0.f, 31.f, a1[i], 0, o1[i]); memcpy(F1.mDescriptors.ptr(i), d1 + 32 * (size_t)i, 32); }
    for (int i = 0; i < n2; ++i) { F2.mvKeysUn[i] = cv::KeyPoint(xy2[2 * i], xy2[2 * i + 1], 31.f, a2[i], 0, o2[i]); memcpy(F2.mDescriptors.ptr(i), d2 + 32 * (size_t)i, 32); }
    std::vector<cv::Point2f> vbPrevMatched(n1);
    for (int i = 0; i < n1; ++i) vbPrevMatched[i] = cv::Point2f(prev[2 * i], prev[2 * i + 1]);
    std::vector<int> vnMatches12;
    ORBmatcher matcher(nnratio, true);
    const int nm = matcher.SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, window);
    const int status = ORBmatcher::LastStatus();
    if (status != 0) fprintf(stderr, "init failed: %s\n", slamit_last_error());
    FILE* f = fopen(argv[3], "wb");
    fwrite(&status, 4, 1, f); fwrite(&nm, 4, 1, f);
    fwrite(vnMatches12.data(), 4, n1, f);
    for (int i = 0; i < n1; ++i) { fwrite(&vbPrevMatched[i].x, 4, 1, f); fwrite(&vbPrevMatched[i].y, 4, 1, f); }
    fclose(f);
    return 0;
}

// ---- SearchByBoW / SearchForTriangulation through the templates ----------------------------------------------------
struct MockBowPoint {
    int id; bool bad;
    bool isBad() const { return bad; }
};
struct MockBowKF {   // doubles as the Frame of SearchByBoW(KeyFrame*, Frame&)
    int N;
    std::map<unsigned, std::vector<unsigned> > mFeatVec;   // the shape of DBoW2::FeatureVector
    cv::Mat mDescriptors, Ow, Rcw, tcw;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<MockBowPoint*> matches;
    std::vector<float> mvuRight, mvScaleFactors, mvLevelSigma2;
    float fx, fy, cx, cy;
    std::vector<MockBowPoint*> GetMapPointMatches() { return matches; }
    MockBowPoint* GetMapPoint(size_t i) { return matches[i]; }
    cv::Mat GetCameraCenter() { return Ow; }
    cv::Mat GetRotation() { return Rcw; }
    cv::Mat GetTranslation() { return tcw; }
};
// problem.bin: int32 variant (0 KeyFrame/Frame, 1 KeyFrame/KeyFrame, 2 triangulation) n1 n2 ; float nnratio ;
//   per side: u8 desc[32 n], float angle[n], int32 node[n], int32 mp[n] (0 none, 1 good, 2 bad), float xy[2 n], int32 octave[n] ;
//   float F12[9] Cw[3] R2w[9] t2w[3] intr[4] scale[8] sigma2[8]
// out.bin: int32 status nmatches ; variant 0: int32[n2] (side-1 index of the map point given to frame feature i, or -1),
//   variant 1: int32[n1] (side-2 index whose map point was matched, or -1), variant 2: int32 npairs, then int32 pairs[2 npairs]
static void fill_side(Reader& R, int n, MockBowKF& K, std::vector<MockBowPoint>& pts) {
    const unsigned char* d = R.arr<unsigned char>(32 * (size_t)n);
    const float* ang = R.arr<float>(n); const int* node = R.arr<int>(n); const int* mp = R.arr<int>(n);
    const float* xy = R.arr<float>(2 * (size_t)n); const int* oct = R.arr<int>(n);
    K.N = n; K.mDescriptors = cv::Mat(n, 32, CV_8U); K.mvKeysUn.resize(n); K.matches.assign(n, (MockBowPoint*)0); K.mvuRight.assign(n, -1.f);
    pts.resize(n);
    for (int i = 0; i < n; ++i) {
        memcpy(K.mDescriptors.ptr(i), d + 32 * (size_t)i, 32);
        K.mvKeysUn[i] = cv::KeyPoint(xy[2 * i], xy[2 * i + 1], 31.f, ang[i], 0, oct[i]);
        K.mFeatVec[(unsigned)node[i]].push_back((unsigned)i);
        pts[i].id = i; pts[i].bad = mp[i] == 2;
        if (mp[i]) K.matches[i] = &pts[i];
    }
    K.mvKeys = K.mvKeysUn;
}
static int run_bow(int argc, char** argv) {
    if (argc < 4) return 2;
    std::vector<unsigned char> raw = slurp(argv[2]);
    Reader R{raw.data()};
    const int variant = R.get<int>(), n1 = R.get<int>(), n2 = R.get<int>();
    const float nnratio = R.get<float>();
    MockBowKF K1, K2;
    std::vector<MockBowPoint> p1, p2;
    fill_side(R, n1, K1, p1);
    fill_side(R, n2, K2, p2);
    const float* F = R.arr<float>(9); const float* Cw = R.arr<float>(3); const float* R2 = R.arr<float>(9); const float* t2 = R.arr<float>(3);
    const float* intr = R.arr<float>(4); const float* scale = R.arr<float>(8); const float* sig = R.arr<float>(8);
    K1.Ow = mat_from(Cw, 3);
    K2.Rcw = cv::Mat(3, 3, CV_32F); K2.tcw = mat_from(t2, 3);
    cv::Mat F12(3, 3, CV_32F);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { K2.Rcw.at<float>(r, c) = R2[3 * r + c]; F12.at<float>(r, c) = F[3 * r + c]; }
    K2.fx = intr[0]; K2.fy = intr[1]; K2.cx = intr[2]; K2.cy = intr[3];
    K2.mvScaleFactors.assign(scale, scale + 8); K2.mvLevelSigma2.assign(sig, sig + 8);
    ORBmatcher matcher(nnratio, true);
    int nm = 0;
    std::vector<int> out;
    if (variant == 0) {
        std::vector<MockBowPoint*> got;
        nm = matcher.SearchByBoW(&K1, K2, got);
        out.assign(n2, -1);
        for (int i = 0; i < n2 && i < (int)got.size(); ++i) if (got[i]) out[i] = got[i]->id;
    } else if (variant == 1) {
        std::vector<MockBowPoint*> got;
        nm = matcher.SearchByBoW(&K1, &K2, got);
        out.assign(n1, -1);
        for (int i = 0; i < n1 && i < (int)got.size(); ++i) if (got[i]) out[i] = got[i]->id;
    } else {
        std::vector<std::pair<size_t, size_t> > pairs;
        nm = matcher.SearchForTriangulation(&K1, &K2, F12, pairs, false);
        out.push_back((int)pairs.size());
        for (size_t k = 0; k < pairs.size(); ++k) { out.push_back((int)pairs[k].first); out.push_back((int)pairs[k].second); }
    }
    const int status = ORBmatcher::LastStatus();
    if (status != 0) fprintf(stderr, "bow failed: %s\n", slamit_last_error());
    FILE* f = fopen(argv[3], "wb");
    fwrite(&status, 4, 1, f); fwrite(&nm, 4, 1, f);
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    return 0;
}

// ---- the Sim3 drivers through the templates -----------------------------------------------------------------------
// problem.bin: int32 variant (0 SearchByProjection(pKF, Scw, ...), 1 Fuse(pKF, Scw, ...), 2 SearchBySim3) nkf ; float th ;
//   map points: int32 m ; float pos[3m] normal[3m] maxd[m] mind[m] ; int32 level[m] bad[m] idx_in_kf2[m] ; u8 desc[32m]
//   per keyframe (nkf of them): float R[9] t[3] intr[4] bounds[6] scale[8] ; int32 n ; float xy[2n] ; int32 octave[n] mp[n] (map point
//     id sitting at the keypoint or -1) ; u8 desc[32n]
//   variant 0/1: float Scw[12] (rows 0..2 of the 4x4) ; variant 0: int32 matched[n] ; variant 2: float s12 R12[9] t12[3] ; int32 matches12[n1]
// out.bin: int32 status count ; variant 0: int32 matched[n] ; variant 1: int32 replace[m] addedAt[m] owner[n] ; variant 2: int32 matches12[n1]
static void read_kf(Reader& Rd, MockFuseKF& KF, std::vector<MockFusePoint>& pts) {
    const float* Rv = Rd.arr<float>(9); const float* tv = Rd.arr<float>(3); const float* intr = Rd.arr<float>(4);
    const float* b = Rd.arr<float>(6); const float* scale = Rd.arr<float>(8);
    const int n = Rd.get<int>();
    const float* xy = Rd.arr<float>(2 * (size_t)n); const int* oct = Rd.arr<int>(n); const int* mp = Rd.arr<int>(n);
    const unsigned char* kd = Rd.arr<unsigned char>(32 * (size_t)n);
    KF.R = cv::Mat(3, 3, CV_32F); KF.t = cv::Mat(3, 1, CV_32F);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) KF.R.at<float>(r, c) = Rv[3 * r + c]; KF.t.at<float>(r, 0) = tv[r]; }
    KF.fx = intr[0]; KF.fy = intr[1]; KF.cx = intr[2]; KF.cy = intr[3]; KF.mfLogScaleFactor = 0.18232f;
    KF.mnMinX = b[0]; KF.mnMaxX = b[1]; KF.mnMinY = b[2]; KF.mnMaxY = b[3]; KF.mfGridElementWidthInv = b[4]; KF.mfGridElementHeightInv = b[5];
    KF.mvScaleFactors.assign(scale, scale + 8);
    KF.mvuRight.assign(n, -1.f); KF.mvKeysUn.resize(n); KF.mps.assign(n, (MockFusePoint*)0);
    KF.mDescriptors = cv::Mat(n, 32, CV_8U);
    for (int i = 0; i < n; ++i) {
        KF.mvKeysUn[i] = cv::KeyPoint(xy[2 * i], xy[2 * i + 1], 31.f, -1.f, 0, oct[i]);
        memcpy(KF.mDescriptors.ptr(i), kd + 32 * (size_t)i, 32);
        if (mp[i] >= 0) KF.mps[i] = &pts[mp[i]];
    }
}
static int run_sim3(int argc, char** argv) {
    if (argc < 4) return 2;
    std::vector<unsigned char> raw = slurp(argv[2]);
    Reader Rd{raw.data()};
    const int variant = Rd.get<int>(), nkf = Rd.get<int>();
    const float th = Rd.get<float>();
    const int m = Rd.get<int>();
    const float* pos = Rd.arr<float>(3 * (size_t)m); const float* nrm = Rd.arr<float>(3 * (size_t)m);
    const float* maxd = Rd.arr<float>(m); const float* mind = Rd.arr<float>(m);
    const int* level = Rd.arr<int>(m); const int* bad = Rd.arr<int>(m); const int* idx2 = Rd.arr<int>(m);
    const unsigned char* md = Rd.arr<unsigned char>(32 * (size_t)m);
    std::vector<MockFusePoint> pts(m);
    for (int j = 0; j < m; ++j) {
        MockFusePoint& p = pts[j];
        p.id = j; p.level = level[j]; p.bad = bad[j] != 0; p.maxd = maxd[j]; p.mind = mind[j]; p.idxInKF2 = idx2[j];
        p.pos = mat_from(pos + 3 * (size_t)j, 3); p.normal = mat_from(nrm + 3 * (size_t)j, 3);
        p.desc = cv::Mat(1, 32, CV_8U); memcpy(p.desc.ptr(0), md + 32 * (size_t)j, 32);
    }
    MockFuseKF KF[2];
    for (int k = 0; k < nkf && k < 2; ++k) read_kf(Rd, KF[k], pts);
    ORBmatcher matcher(0.75f, true);
    std::vector<int> out;
    int count = 0;
    if (variant == 0 || variant == 1) {
        const float* S = Rd.arr<float>(12);
        cv::Mat Scw = cv::Mat::zeros(4, 4, CV_32F);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) Scw.at<float>(r, c) = S[4 * r + c];
        Scw.at<float>(3, 3) = 1.f;
        std::vector<MockFusePoint*> vp(m);
        for (int j = 0; j < m; ++j) vp[j] = &pts[j];
        const int n = (int)KF[0].mvKeysUn.size();
        if (variant == 0) {
            const int* init = Rd.arr<int>(n);
            std::vector<MockFusePoint*> vpMatched(n, (MockFusePoint*)0);
            for (int i = 0; i < n; ++i) if (init[i] >= 0) vpMatched[i] = &pts[init[i]];
            count = matcher.SearchByProjection(&KF[0], Scw, vp, vpMatched, (int)th);
            for (int i = 0; i < n; ++i) out.push_back(vpMatched[i] ? vpMatched[i]->id : -1);
        } else {
            std::vector<MockFusePoint*> vpReplace(m, (MockFusePoint*)0);
            count = matcher.Fuse(&KF[0], Scw, vp, th, vpReplace);
            for (int j = 0; j < m; ++j) out.push_back(vpReplace[j] ? vpReplace[j]->id : -1);
            for (int j = 0; j < m; ++j) out.push_back(pts[j].addedAt);
            for (int i = 0; i < n; ++i) out.push_back(KF[0].mps[i] ? KF[0].mps[i]->id : -1);
        }
    } else {
        const float s12 = Rd.get<float>();
        const float* Rv = Rd.arr<float>(9); const float* tv = Rd.arr<float>(3);
        cv::Mat R12(3, 3, CV_32F), t12 = mat_from(tv, 3);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R12.at<float>(r, c) = Rv[3 * r + c];
        const int n1 = (int)KF[0].mvKeysUn.size();
        const int* init = Rd.arr<int>(n1);
        std::vector<MockFusePoint*> vpMatches12(n1, (MockFusePoint*)0);
        for (int i = 0; i < n1; ++i) if (init[i] >= 0) vpMatches12[i] = &pts[init[i]];
        count = matcher.SearchBySim3(&KF[0], &KF[1], vpMatches12, s12, R12, t12, th);
        for (int i = 0; i < n1; ++i) out.push_back(vpMatches12[i] ? vpMatches12[i]->id : -1);
    }
    const int status = ORBmatcher::LastStatus();
    if (status != 0) fprintf(stderr, "sim3 failed: %s\n", slamit_last_error());
    FILE* f = fopen(argv[3], "wb");
    fwrite(&status, 4, 1, f); fwrite(&count, 4, 1, f);
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    return 0;
}

// ---- Optimizer::OptimizeSim3 through the template ----------------------------------------------------------------
struct MockSim3 { double R[9], t[3], s; };
inline void slamit_shim_sim3_get(const MockSim3& S, double R[9], double t[3], double& s) { memcpy(R, S.R, 72); memcpy(t, S.t, 24); s = S.s; }
inline void slamit_shim_sim3_set(MockSim3& S, const double R[9], const double t[3], double s) { memcpy(S.R, R, 72); memcpy(S.t, t, 24); S.s = s; }
// problem.bin: int32 n1 n2 m fix_scale ; float th2 ; double S12[13] (R, t, s) ; per keyframe: float K[4] (fx fy cx cy) R[9] t[3] invsig[8],
//   keypoints float xy[2n] int32 octave[n] int32 mp[n] (own map point id or -1) ; map points: float pos[3m], int32 bad[m] idx_in_kf2[m] ;
//   int32 matches1[n1] (map point id of the KF2 point matched to keypoint i of KF1, or -1)
// out.bin: int32 status n_inliers ; int32 matches1[n1] ; double S12[13]
static int run_osim3(int argc, char** argv) {
    if (argc < 4) return 2;
    std::vector<unsigned char> raw = slurp(argv[2]);
    Reader Rd{raw.data()};
    const int n[2] = {Rd.get<int>(), Rd.get<int>()};
    const int m = Rd.get<int>(), fix = Rd.get<int>();
    const float th2 = Rd.get<float>();
    MockSim3 S12;
    { const double* v = Rd.arr<double>(13); memcpy(S12.R, v, 72); memcpy(S12.t, v + 9, 24); S12.s = v[12]; }
    MockFuseKF KF[2];
    const int* mpk[2];
    for (int k = 0; k < 2; ++k) {
        const float* K = Rd.arr<float>(4); const float* Rv = Rd.arr<float>(9); const float* tv = Rd.arr<float>(3); const float* isg = Rd.arr<float>(8);
        const float* xy = Rd.arr<float>(2 * (size_t)n[k]); const int* oct = Rd.arr<int>(n[k]); mpk[k] = Rd.arr<int>(n[k]);
        KF[k].mK = cv::Mat::zeros(3, 3, CV_32F);
        KF[k].mK.at<float>(0, 0) = K[0]; KF[k].mK.at<float>(1, 1) = K[1]; KF[k].mK.at<float>(0, 2) = K[2]; KF[k].mK.at<float>(1, 2) = K[3]; KF[k].mK.at<float>(2, 2) = 1.f;
        KF[k].R = cv::Mat(3, 3, CV_32F); KF[k].t = mat_from(tv, 3);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) KF[k].R.at<float>(r, c) = Rv[3 * r + c];
        KF[k].mvInvLevelSigma2.assign(isg, isg + 8);
        KF[k].mvKeysUn.resize(n[k]);
        for (int i = 0; i < n[k]; ++i) KF[k].mvKeysUn[i] = cv::KeyPoint(xy[2 * i], xy[2 * i + 1], 31.f, -1.f, 0, oct[i]);
    }
    const float* pos = Rd.arr<float>(3 * (size_t)m); const int* bad = Rd.arr<int>(m); const int* idx2 = Rd.arr<int>(m);
    std::vector<MockFusePoint> pts(m);
    for (int j = 0; j < m; ++j) { pts[j].id = j; pts[j].bad = bad[j] != 0; pts[j].idxInKF2 = idx2[j]; pts[j].pos = mat_from(pos + 3 * (size_t)j, 3); }
    for (int k = 0; k < 2; ++k) { KF[k].mps.assign(n[k], (MockFusePoint*)0); for (int i = 0; i < n[k]; ++i) if (mpk[k][i] >= 0) KF[k].mps[i] = &pts[mpk[k][i]]; }
    const int* m1 = Rd.arr<int>(n[0]);
    std::vector<MockFusePoint*> vpMatches1(n[0], (MockFusePoint*)0);
    for (int i = 0; i < n[0]; ++i) if (m1[i] >= 0) vpMatches1[i] = &pts[m1[i]];
    const int nin = Optimizer::OptimizeSim3(&KF[0], &KF[1], vpMatches1, S12, th2, fix != 0);
    const int status = Optimizer::LastStatus();
    if (status != 0) fprintf(stderr, "osim3 failed: %s\n", slamit_last_error());
    FILE* f = fopen(argv[3], "wb");
    fwrite(&status, 4, 1, f); fwrite(&nin, 4, 1, f);
    for (int i = 0; i < n[0]; ++i) { int id = vpMatches1[i] ? vpMatches1[i]->id : -1; fwrite(&id, 4, 1, f); }
    fwrite(S12.R, 8, 9, f); fwrite(S12.t, 8, 3, f); fwrite(&S12.s, 8, 1, f);
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
    if (mode == "search") return run_search(argc, argv);
    if (mode == "frame") return run_frame(argc, argv);
    if (mode == "fuse") return run_fuse(argc, argv);
    if (mode == "init") return run_init(argc, argv);
    if (mode == "bow") return run_bow(argc, argv);
    if (mode == "sim3") return run_sim3(argc, argv);
    if (mode == "osim3") return run_osim3(argc, argv);
    return 2;
}
