// ORBextractor.cc — see ORBextractor.h.  Host adapter only: owns a slamit_orb handle per frame
// geometry and converts between cv:: containers and the C-ABI's POD buffers.
#include "ORBextractor.h"

#include <string.h>

#include "../../include/slamit.h"

namespace ORB_SLAM2 {

static_assert(sizeof(cv::KeyPoint) == sizeof(slamit_kp), "slamit_kp must mirror cv::KeyPoint");

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST),
      minThFAST(_minThFAST), handle(0), boundW(-1), boundH(-1), device(0), lastStatus(0), exportPyramid(false) {
    // The getters must answer before the first frame (Frame's constructors read them), so the
    // tables come from a geometry-less bind: the library computes them from the parameters alone.
    mvImagePyramid.resize(nlevels);
    mvPaddedPyramid.resize(nlevels);
    bind(640, 480);
}

ORBextractor::~ORBextractor() { slamit_orb_destroy(handle); }

void ORBextractor::SetDevice(int d) {
    if (d != device) {
        device = d;
        slamit_orb_destroy(handle);
        handle = 0;
        boundW = boundH = -1;
    }
}

const char* ORBextractor::lastError() const { return slamit_last_error(); }

bool ORBextractor::bind(int width, int height) {
    if (handle && width == boundW && height == boundH) return true;
    slamit_orb_destroy(handle);
    handle = 0;
    slamit_orb_params p;
    p.nfeatures = nfeatures; p.scale_factor = (float)scaleFactor; p.nlevels = nlevels;
    p.ini_th_fast = iniThFAST; p.min_th_fast = minThFAST; p.width = width; p.height = height; p.max_batch = 1;
    lastStatus = slamit_orb_create(&p, device, &handle);
    if (lastStatus != SLAMIT_OK) { handle = 0; boundW = boundH = -1; return false; }
    boundW = width; boundH = height;
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels); mnFeaturesPerLevel.resize(nlevels);
    slamit_orb_tables(handle, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                      mvInvLevelSigma2.data(), mnFeaturesPerLevel.data());
    return true;
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                              cv::OutputArray _descriptors) {
    if (_image.empty()) return;  // same silent return as the reference
    cv::Mat image = _image.getMat();
    if (image.type() != CV_8UC1) { lastStatus = SLAMIT_ERR_ARG; _keypoints.clear(); _descriptors.release(); return; }
    if (!bind(image.cols, image.rows)) { _keypoints.clear(); _descriptors.release(); return; }
    const int cap = slamit_orb_max_keypoints(handle);
    std::vector<slamit_kp> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * SLAMIT_DESC_BYTES);
    int n = 0;
    lastStatus = slamit_orb_extract(handle, image.ptr<uint8_t>(0), image.step, kps.data(), desc.data(), cap, &n);
    if (lastStatus != SLAMIT_OK) n = 0;
    if (n == 0) {
        _descriptors.release();
    } else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        for (int i = 0; i < n; ++i) memcpy(d.ptr(i), &desc[(size_t)i * SLAMIT_DESC_BYTES], SLAMIT_DESC_BYTES);
    }
    _keypoints.resize(n);
    if (n) memcpy(static_cast<void*>(&_keypoints[0]), kps.data(), sizeof(slamit_kp) * (size_t)n);
    if (exportPyramid && lastStatus == SLAMIT_OK) {
        for (int l = 0; l < nlevels; ++l) {
            int w = 0, h = 0;
            if (slamit_orb_level(handle, 0, l, 0, 0, &w, &h) != SLAMIT_OK) break;
            cv::Mat padded(h + 2 * SLAMIT_EDGE_THRESHOLD, w + 2 * SLAMIT_EDGE_THRESHOLD, CV_8UC1);
            if (slamit_orb_level(handle, 0, l, padded.data, (size_t)padded.rows * padded.step, &w, &h) != SLAMIT_OK) break;
            // the reference's mvImagePyramid[l] is the ROI inside the padded buffer
            mvPaddedPyramid[l] = padded;  // owns the bytes; the view below points into it
            mvImagePyramid[l] = cv::Mat(h, w, CV_8UC1, padded.data + (size_t)SLAMIT_EDGE_THRESHOLD * padded.step + SLAMIT_EDGE_THRESHOLD,
                                        padded.step);
        }
    }
}

}  // namespace ORB_SLAM2
