// ORBmatcher.cc — see ORBmatcher.h.  Distances and selection run in libslamit_hip.so; the
// acceptance thresholds and the rotation histogram are host logic as in the reference.
#include "ORBmatcher.h"

#include <math.h>

#include "../../include/slamit.h"

namespace ORB_SLAM2 {

const int ORBmatcher::TH_HIGH = 100;     // ORBmatcher.cc:37
const int ORBmatcher::TH_LOW = 50;       // :38
const int ORBmatcher::HISTO_LENGTH = 30; // :39

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

namespace {
// descriptor tables may be row-padded cv::Mat views; the C-ABI wants 32-byte rows
const uint8_t* packed_rows(const cv::Mat& m, std::vector<uint8_t>& tmp) {
    if (m.rows == 0) return 0;
    if (m.isContinuous() && m.cols == SLAMIT_DESC_BYTES) return m.ptr<uint8_t>(0);
    tmp.resize((size_t)m.rows * SLAMIT_DESC_BYTES);
    for (int r = 0; r < m.rows; ++r) memcpy(&tmp[(size_t)r * SLAMIT_DESC_BYTES], m.ptr<uint8_t>(r), SLAMIT_DESC_BYTES);
    return tmp.data();
}
}  // namespace

int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
    unsigned short d = 0;
    if (slamit_hamming_matrix(a.ptr<uint8_t>(0), 1, b.ptr<uint8_t>(0), 1, &d) != SLAMIT_OK) return 256;
    return d;
}

bool ORBmatcher::DistanceMatrix(const cv::Mat& query, const cv::Mat& train, std::vector<unsigned short>& dist) {
    std::vector<uint8_t> tq, tt;
    dist.assign((size_t)query.rows * train.rows, 0);
    return slamit_hamming_matrix(packed_rows(query, tq), query.rows, packed_rows(train, tt), train.rows, dist.data()) == SLAMIT_OK;
}

bool ORBmatcher::BestTwo(const cv::Mat& query, const cv::Mat& train, std::vector<int>& bestIdx,
                         std::vector<int>& bestDist, std::vector<int>& secondDist) {
    std::vector<uint8_t> tq, tt;
    bestIdx.assign(query.rows, -1); bestDist.assign(query.rows, 256); secondDist.assign(query.rows, 256);
    if (query.rows == 0) return true;
    return slamit_hamming_best2(packed_rows(query, tq), query.rows, packed_rows(train, tt), train.rows, bestIdx.data(),
                                bestDist.data(), secondDist.data()) == SLAMIT_OK;
}

static int g_status = 0;
int ORBmatcher::LastStatus() { return g_status; }
void ORBmatcher::setStatus(int rc) { g_status = rc; }

bool ORBmatcher::GuidedSearch(const std::vector<cv::KeyPoint>& keysUn, const cv::Mat& descriptors,
                              const std::vector<uint8_t>& kpTaken, float minX, float minY, float invW, float invH,
                              const GuidedQueries& q, int thDist, bool useRatio, float nnratio, std::vector<int>& matchKp,
                              float chi2Gate, const std::vector<float>* invLevelSigma2, int mode, std::vector<int>* acceptedKp) {
    const int n = (int)keysUn.size(), m = q.size();
    matchKp.assign(m, -1);
    g_status = SLAMIT_OK;
    if (m == 0) return true;
    std::vector<float> xy(2 * (size_t)n);
    std::vector<int32_t> oct(n);
    for (int i = 0; i < n; ++i) { xy[2 * i] = keysUn[i].pt.x; xy[2 * i + 1] = keysUn[i].pt.y; oct[i] = keysUn[i].octave; }
    std::vector<uint8_t> td;
    slamit_frame_view fv;
    fv.n = n; fv.kp_xy = xy.data(); fv.kp_octave = oct.data(); fv.desc = n ? packed_rows(descriptors, td) : nullptr;
    fv.kp_taken = kpTaken.data(); fv.min_x = minX; fv.min_y = minY; fv.inv_w = invW; fv.inv_h = invH;
    slamit_search_queries sq;
    sq.m = m; sq.uvr = q.uvr.data(); sq.level_min = q.lmin.data(); sq.level_max = q.lmax.data(); sq.desc = q.desc.data();
    sq.valid = q.valid.data(); sq.takes = q.takes.data();
    slamit_search_rule rule;
    rule.th_dist = thDist; rule.use_ratio = useRatio ? 1 : 0; rule.nnratio = nnratio;
    rule.chi2_gate = chi2Gate;
    rule.mode = mode;
    for (int i = 0; i < 16; ++i) rule.inv_level_sigma2[i] = (invLevelSigma2 && i < (int)invLevelSigma2->size()) ? (*invLevelSigma2)[i] : 1.f;
    int nm = 0;
    if (acceptedKp) acceptedKp->assign(m, -1);
    g_status = slamit_guided_search(0, &fv, &sq, &rule, matchKp.data(), &nm, nullptr, acceptedKp ? acceptedKp->data() : nullptr, nullptr, nullptr);
    return g_status == SLAMIT_OK;
}

bool ORBmatcher::BowSearch(const cv::Mat& desc1, const std::vector<uint8_t>& valid1, const cv::Mat& desc2, const std::vector<uint8_t>* valid2,
                           const BowGroups& g, int th, bool thInclusive, float nnratio, const EpipolarGate* gate, std::vector<int>& match12) {
    const int n1 = desc1.rows, n2 = desc2.rows;
    match12.assign(n1, -1);
    g_status = SLAMIT_OK;
    if (n1 == 0 || n2 == 0 || g.size() == 0) return true;
    std::vector<uint8_t> t1, t2;
    slamit_bow_groups gg;
    gg.n_groups = g.size(); gg.q_ptr = g.q_ptr.data(); gg.q_idx = g.q_idx.data(); gg.c_ptr = g.c_ptr.data(); gg.c_idx = g.c_idx.data();
    slamit_bow_rule rule;
    memset(&rule, 0, sizeof(rule));
    rule.mode = gate ? 1 : 0; rule.th = th; rule.th_inclusive = thInclusive ? 1 : 0; rule.nnratio = nnratio;
    std::vector<float> xy1, xy2;
    std::vector<int32_t> oct2;
    if (gate) {
        memcpy(rule.F12, gate->F12, sizeof(rule.F12)); rule.ex = gate->ex; rule.ey = gate->ey;
        xy1.resize(2 * (size_t)n1); xy2.resize(2 * (size_t)n2); oct2.resize(n2);
        for (int i = 0; i < n1; ++i) { xy1[2 * i] = (*gate->keys1)[i].pt.x; xy1[2 * i + 1] = (*gate->keys1)[i].pt.y; }
        for (int i = 0; i < n2; ++i) { xy2[2 * i] = (*gate->keys2)[i].pt.x; xy2[2 * i + 1] = (*gate->keys2)[i].pt.y; oct2[i] = (*gate->keys2)[i].octave; }
        rule.kp1_xy = xy1.data(); rule.kp2_xy = xy2.data(); rule.kp2_octave = oct2.data();
        for (int i = 0; i < 16; ++i) {
            rule.scale_factor[i] = i < (int)gate->scaleFactors->size() ? (*gate->scaleFactors)[i] : 1.f;
            rule.level_sigma2[i] = i < (int)gate->levelSigma2->size() ? (*gate->levelSigma2)[i] : 1.f;
        }
    }
    int nm = 0;
    g_status = slamit_bow_search(0, packed_rows(desc1, t1), n1, valid1.empty() ? nullptr : valid1.data(), packed_rows(desc2, t2), n2,
                                 valid2 ? valid2->data() : nullptr, &gg, &rule, match12.data(), nullptr, &nm);
    return g_status == SLAMIT_OK;
}

int ORBmatcher::SearchBruteForce(const std::vector<cv::KeyPoint>& keys1, const cv::Mat& desc1,
                                 const std::vector<cv::KeyPoint>& keys2, const cv::Mat& desc2,
                                 std::vector<int>& vnMatches12, int th) {
    if (th < 0) th = TH_LOW;
    std::vector<int> idx, best, second;
    vnMatches12.assign(desc1.rows, -1);
    if (!BestTwo(desc1, desc2, idx, best, second)) return 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    for (int i = 0; i < desc1.rows; ++i) {
        if (best[i] <= th && (float)best[i] < mfNNratio * (float)second[i]) {
            vnMatches12[i] = idx[i];
            ++nmatches;
            if (mbCheckOrientation && i < (int)keys1.size() && idx[i] < (int)keys2.size()) {
                float rot = keys1[i].angle - keys2[idx[i]].angle;
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(i);
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; ++b) {
            if (b == ind1 || b == ind2 || b == ind3) continue;
            for (size_t j = 0; j < rotHist[b].size(); ++j) {
                if (vnMatches12[rotHist[b][j]] >= 0) { vnMatches12[rotHist[b][j]] = -1; --nmatches; }
            }
        }
    }
    return nmatches;
}

// three most populated bins; the 2nd/3rd are dropped when below 10 % of the first (ORBmatcher.cc:1605-1646)
void ORBmatcher::ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
    int top[3] = {0, 0, 0};
    int at[3] = {-1, -1, -1};
    for (int i = 0; i < L; ++i) {
        const int s = (int)histo[i].size();
        int pos = s > top[0] ? 0 : s > top[1] ? 1 : s > top[2] ? 2 : 3;
        for (int k = 2; k > pos; --k) { top[k] = top[k - 1]; at[k] = at[k - 1]; }
        if (pos < 3) { top[pos] = s; at[pos] = i; }
    }
    if (top[1] < 0.1f * (float)top[0]) { at[1] = -1; at[2] = -1; }
    else if (top[2] < 0.1f * (float)top[0]) at[2] = -1;
    ind1 = at[0]; ind2 = at[1]; ind3 = at[2];
}

}  // namespace ORB_SLAM2
