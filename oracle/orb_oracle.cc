// orb_oracle.cc — CPU ORACLE for the ORB half of the hot path.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
// library (oracle/liborb_oracle.so).  Nothing under weiner_slamit_v2_amd/ links or calls it.
//
// What it is: a from-scratch, single-threaded, strict-IEEE (-ffp-contract=off, no fast-math)
// restatement of ORB_SLAM2::ORBextractor (reference: oRB_SLAM2_Android/src/main/jni/ORB_SLAM2/
// src/ORBextractor.cc, "S/" below) and of the five OpenCV 2.4.9 primitives it calls
// (cv::resize 8UC1 INTER_LINEAR, cv::copyMakeBorder REFLECT_101, cv::FAST 9/16 + NMS,
// cv::GaussianBlur 7x7 sigma 2 8U, cv::fastAtan2, cvRound), plus ORBmatcher::DescriptorDistance
// with the reference's best/second-best selection rule.
//
// PARITY STATUS: *** parity unpinned *** for the OpenCV primitives.  OpenCV 2.4.9 (Android
// SDK, native) is an un-vendored dependency of the reference (jni/Android.mk:30); neither its
// source nor a binary exists in /root/reference or on this machine, and the reference holds no
// test, fixture or golden vector for this path.  The primitives are
// restated from the published OpenCV 2.4 algorithms (generic C code paths) and pinned only by
// definitional known-answer tests (tests/test_oracle_orb.py): brute-force FAST-9/16 definition,
// an independent exact-integer numpy model of the bilinear resize and the 7x7 blur, atan2 error
// bound, umax table, feature quotas.  The ORBextractor.cc logic itself (cells, thresholds,
// octree, orientation, descriptor, ordering) follows the cited lines one to one.
//
// One deliberate definition: DistributeOctTree sorts
// pair<int, ExtractorNode*> (S/ORBextractor.cc:697), so equal-sized nodes are ordered by heap
// address in the reference.  Here the pointer is replaced by the node's creation sequence
// number (what a bump allocator would produce).
//
// Build: oracle/Makefile (g++ -O2 -std=c++11 -ffp-contract=off).

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <limits.h>

#include <algorithm>
#include <list>
#include <utility>
#include <vector>

#include "../include/slamit.h"
#include "../include/slamit_orb_pattern.h"

namespace {

// ---- OpenCV scalar helpers (Appendix A.6) -------------------------------------------------
// cvRound(double) == lrint == round-half-to-even in the default rounding mode
// (openCVLibrary341/src/sdk/native/jni/include/opencv2/core/fast_math.hpp:118-123 for GCC).
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(double v) { return (int)floor(v); }
inline int cv_ceil(double v) { return (int)ceil(v); }

const int PATCH_SIZE = 31;       // S/ORBextractor.cc:77
const int HALF_PATCH_SIZE = 15;  // :78
const int EDGE_THRESHOLD = 19;   // :79

// BORDER_REFLECT_101 index map (cv::borderInterpolate): gfedcb|abcdefgh|gfedcba
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

struct Plane {  // a padded pyramid level: (w+38) x (h+38), ROI origin at (19,19)
    int w, h, stride;
    std::vector<uint8_t> buf;
    uint8_t* roi() { return &buf[(size_t)EDGE_THRESHOLD * stride + EDGE_THRESHOLD]; }
    const uint8_t* roi() const { return &buf[(size_t)EDGE_THRESHOLD * stride + EDGE_THRESHOLD]; }
};

// cv::copyMakeBorder(src, dst, 19,19,19,19, BORDER_REFLECT_101 [| BORDER_ISOLATED])
// S/ORBextractor.cc:1159-1164.  The ROI of `p` already holds the image; fill the frame.
void fill_border(Plane& p) {
    uint8_t* r = p.roi();
    const int B = EDGE_THRESHOLD;
    for (int y = -B; y < p.h + B; ++y) {
        int sy = reflect101(y, p.h);
        uint8_t* drow = r + (ptrdiff_t)y * p.stride;
        const uint8_t* srow = r + (ptrdiff_t)sy * p.stride;
        if (y < 0 || y >= p.h)
            for (int x = 0; x < p.w; ++x) drow[x] = srow[x];
        for (int x = -B; x < 0; ++x) drow[x] = srow[reflect101(x, p.w)];
        for (int x = p.w; x < p.w + B; ++x) drow[x] = srow[reflect101(x, p.w)];
    }
}

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1, OpenCV 2.4 generic C path:
// fixed-point coefficients (11 bits), horizontal pass to int32, vertical pass
// ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.     S/ORBextractor.cc:1157, Appendix A.2
inline short sat_short_from_float(float v) {
    int iv = cv_round((double)v);
    return (short)(iv < -32768 ? -32768 : iv > 32767 ? 32767 : iv);
}

void resize_linear_8u(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw,
                      int dh, int dstride) {
    const float COEF = 2048.f;  // INTER_RESIZE_COEF_SCALE
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_short_from_float((1.f - fx) * COEF);
        ialpha[dx * 2 + 1] = sat_short_from_float(fx * COEF);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = sat_short_from_float((1.f - fy) * COEF);
        ibeta[dy * 2 + 1] = sat_short_from_float(fy * COEF);
    }
    std::vector<int> row0(dw), row1(dw);
    for (int dy = 0; dy < dh; ++dy) {
        int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);      // row indices are clamped,
        int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);  // the beta weights are not
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sx;  // weight is 0 there (fx forced to 0)
            row0[dx] = S0[sx] * ialpha[dx * 2] + S0[sx1] * ialpha[dx * 2 + 1];
            row1[dx] = S1[sx] * ialpha[dx * 2] + S1[sx1] * ialpha[dx * 2 + 1];
        }
        int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

// cv::getGaussianKernel(7, 2, CV_32F) converted to fixed point with 8 bits
// (createSeparableLinearFilter for 8U->8U smooth symmetric kernels).  Appendix A.4.
void gauss7_taps(int taps[7]) {
    const int n = 7;
    const double sigma = 2.0;
    double scale2X = -0.5 / (sigma * sigma);
    float cf[7];
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        double x = i - (n - 1) * 0.5;
        cf[i] = (float)exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) {
        cf[i] = (float)(cf[i] * sum);
        taps[i] = cv_round((double)(cf[i] * 256.f));  // convertTo(CV_32S, 256)
    }
}

// cv::GaussianBlur(m, m, Size(7,7), 2, 2, BORDER_REFLECT_101) on an un-padded w x h 8UC1 image
// (the reference blurs mvImagePyramid[level].clone(), S/ORBextractor.cc:1116-1117).
// Row pass: int32 sum of taps*src; column pass: (sum + 2^15) >> 16, saturated to uchar.
void gauss7x7_8u(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    int k[7];
    gauss7_taps(k);
    std::vector<int> rows((size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src + (size_t)y * sstride;
        int* r = &rows[(size_t)y * w];
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int i = 0; i < 7; ++i) acc += k[i] * s[reflect101(x + i - 3, w)];
            r[x] = acc;
        }
    }
    for (int y = 0; y < h; ++y) {
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int i = 0; i < 7; ++i) acc += k[i] * rows[(size_t)reflect101(y + i - 3, h) * w + x];
            int v = (acc + (1 << 15)) >> 16;
            d[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

// cv::fastAtan2 (OpenCV 2.4): 7th-order odd polynomial, degrees.  Appendix A.5.
float fast_atan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---- cv::FAST(img, kps, threshold, nonmaxSuppression=true), FAST_t<16> ----------------------
// Appendix A.1.  (x, y) offsets of the 16-pixel Bresenham circle of radius 3.
const int RING[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                         {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

struct FastKp { int x, y, score; };

int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
    const int N = 25;
    int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; ++k) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        for (int j = 4; j <= 8; ++j) a = std::min(a, (int)d[k + j]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        for (int j = 3; j <= 5; ++j) b = std::max(b, (int)d[k + j]);
        if (b >= b0) continue;
        for (int j = 6; j <= 8; ++j) b = std::max(b, (int)d[k + j]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

void fast9_16(const uint8_t* img, int cols, int rows, int step, int threshold,
              std::vector<FastKp>& out) {
    out.clear();
    if (cols < 7 || rows < 7) return;
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; ++k) pixel[k] = RING[k][0] + RING[k][1] * step;
    for (int k = 16; k < N; ++k) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    // three rolling rows of scores (uchar) and of corner positions
    std::vector<uint8_t> sbuf((size_t)cols * 3, 0);
    std::vector<int> cbuf((size_t)(cols + 1) * 3, 0);
    uint8_t* buf[3] = {&sbuf[0], &sbuf[cols], &sbuf[2 * (size_t)cols]};
    int* cpbuf[3] = {&cbuf[1], &cbuf[cols + 2], &cbuf[2 * (size_t)cols + 3]};
    for (int i = 3; i < rows - 2; ++i) {
        const uint8_t* ptr = img + (size_t)i * step + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3];
        memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; ++j, ++ptr) {
                int v = ptr[0];
                int lo = v - threshold, hi = v + threshold;
                // quick reject (pure optimisation, as in cv::FAST): every 9-arc contains one
                // pixel of each diametral pair, so some class (darker=1 / brighter=2) must be
                // present in all 8 pairs
                int d = 3;
                for (int k = 0; k < 8 && d; ++k) {
                    int p = ptr[pixel[k]], q = ptr[pixel[k + 8]];
                    d &= (p < lo ? 1 : p > hi ? 2 : 0) | (q < lo ? 1 : q > hi ? 2 : 0);
                }
                if (!d) continue;
                // >= 9 contiguous ring pixels darker than v-t, or brighter than v+t,
                // over the ring extended to 25 entries
                int cd = 0, cb = 0;
                bool corner = false;
                for (int k = 0; k < N && !corner; ++k) {
                    int x = ptr[pixel[k]];
                    if (x < lo) { if (++cd > K) corner = true; } else cd = 0;
                    if (x > hi) { if (++cb > K) corner = true; } else cb = 0;
                }
                if (corner) {
                    cornerpos[ncorners++] = j;
                    curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; ++k) {
            int j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] &&
                score > pprev[j] && score > pprev[j + 1] && score > curr[j - 1] &&
                score > curr[j] && score > curr[j + 1]) {
                FastKp kp = {j, i - 1, score};
                out.push_back(kp);
            }
        }
    }
}

// ---- ORBextractor ---------------------------------------------------------------------------

struct Key {  // cv::KeyPoint subset used while distributing
    float x, y, response;
};

struct Node {  // ExtractorNode, include/ORBextractor.h:32-44
    std::vector<Key> keys;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::list<Node>::iterator lit;
    bool noMore;
    long seq;  // creation sequence number: replaces the heap address in the tie-break
    Node() : noMore(false), seq(0) {}
};

// ExtractorNode::DivideNode, S/ORBextractor.cc:494-550
void divide_node(const Node& n, Node& n1, Node& n2, Node& n3, Node& n4) {
    const int halfX = (int)ceil(static_cast<float>(n.URx - n.ULx) / 2);
    const int halfY = (int)ceil(static_cast<float>(n.BRy - n.ULy) / 2);
    n1.ULx = n.ULx; n1.ULy = n.ULy;
    n1.URx = n.ULx + halfX; n1.URy = n.ULy;
    n1.BLx = n.ULx; n1.BLy = n.ULy + halfY;
    n1.BRx = n.ULx + halfX; n1.BRy = n.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = n.URx; n2.URy = n.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = n.URx; n2.BRy = n.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = n.BLx; n3.BLy = n.BLy;
    n3.BRx = n1.BRx; n3.BRy = n.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = n.BRx; n4.BRy = n.BRy;
    for (size_t i = 0; i < n.keys.size(); ++i) {
        const Key& kp = n.keys[i];
        if (kp.x < n1.URx) {
            if (kp.y < n1.BRy) n1.keys.push_back(kp);
            else n3.keys.push_back(kp);
        } else if (kp.y < n1.BRy) n2.keys.push_back(kp);
        else n4.keys.push_back(kp);
    }
    if (n1.keys.size() == 1) n1.noMore = true;
    if (n2.keys.size() == 1) n2.noMore = true;
    if (n3.keys.size() == 1) n3.noMore = true;
    if (n4.keys.size() == 1) n4.noMore = true;
}

typedef std::pair<int, std::pair<long, Node*> > SizeSeqNode;  // (size, (seq, node))

// children with keys are pushed to the FRONT in the order n1..n4 (S/ORBextractor.cc:634-673)
inline void push_children(std::list<Node>& lNodes, Node* ch[4], long& seq, int* nToExpand,
                          std::vector<SizeSeqNode>& vSize) {
    for (int c = 0; c < 4; ++c) {
        if (ch[c]->keys.size() > 0) {
            ch[c]->seq = seq++;
            lNodes.push_front(*ch[c]);
            if (ch[c]->keys.size() > 1) {
                if (nToExpand) ++*nToExpand;
                vSize.push_back(std::make_pair((int)ch[c]->keys.size(),
                                               std::make_pair(lNodes.front().seq, &lNodes.front())));
                lNodes.front().lit = lNodes.begin();
            }
        }
    }
}

// ORBextractor::DistributeOctTree, S/ORBextractor.cc:552-776
std::vector<Key> distribute_octree(const std::vector<Key>& toDistribute, int minX, int maxX,
                                   int minY, int maxY, int N) {
    const int nIni = (int)round(static_cast<float>(maxX - minX) / (maxY - minY));
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<Node> lNodes;
    std::vector<Node*> vpIniNodes(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; ++i) {
        Node ni;
        ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
        ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.seq = seq++;
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < toDistribute.size(); ++i) {
        const Key& kp = toDistribute[i];
        vpIniNodes[(int)(kp.x / hX)]->keys.push_back(kp);
    }
    std::list<Node>::iterator lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->keys.size() == 1) { lit->noMore = true; ++lit; }
        else if (lit->keys.empty()) lit = lNodes.erase(lit);
        else ++lit;
    }
    bool bFinish = false;
    std::vector<SizeSeqNode> vSizeAndPointerToNode;
    while (!bFinish) {
        int prevSize = (int)lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();
        while (lit != lNodes.end()) {
            if (lit->noMore) { ++lit; continue; }
            Node n1, n2, n3, n4;
            divide_node(*lit, n1, n2, n3, n4);
            Node* ch[4] = {&n1, &n2, &n3, &n4};
            push_children(lNodes, ch, seq, &nToExpand, vSizeAndPointerToNode);
            lit = lNodes.erase(lit);
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = (int)lNodes.size();
                std::vector<SizeSeqNode> vPrev = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();
                std::sort(vPrev.begin(), vPrev.end());  // (size, creation seq) ascending
                for (int j = (int)vPrev.size() - 1; j >= 0; --j) {
                    Node n1, n2, n3, n4;
                    Node* parent = vPrev[j].second.second;
                    divide_node(*parent, n1, n2, n3, n4);
                    Node* ch[4] = {&n1, &n2, &n3, &n4};
                    push_children(lNodes, ch, seq, 0, vSizeAndPointerToNode);
                    lNodes.erase(parent->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }
    // best response per node, first maximum wins (S/ORBextractor.cc:755-773)
    std::vector<Key> result;
    for (lit = lNodes.begin(); lit != lNodes.end(); ++lit) {
        const std::vector<Key>& v = lit->keys;
        const Key* best = &v[0];
        float maxResponse = best->response;
        for (size_t k = 1; k < v.size(); ++k)
            if (v[k].response > maxResponse) { best = &v[k]; maxResponse = v[k].response; }
        result.push_back(*best);
    }
    return result;
}

struct Oracle {
    // parameters (S/ORBextractor.cc:415-420); scaleFactor is a double member holding a float
    int nfeatures, nlevels, iniThFAST, minThFAST;
    double scaleFactor;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    // state of the last run
    std::vector<Plane> pyramid;
    std::vector<std::vector<uint8_t> > blurred;       // w*h per level (only levels with keypoints)
    std::vector<std::vector<int32_t> > candidates;    // x,y,score triplets per level
    std::vector<std::vector<slamit_kp> > levelKps;    // after octree + orientation (level coords)
    std::vector<slamit_kp> kps;
    std::vector<uint8_t> desc;
};

// ORBextractor::ORBextractor, S/ORBextractor.cc:415-482
void oracle_init(Oracle& o, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    o.nfeatures = nfeatures; o.nlevels = nlevels; o.iniThFAST = iniTh; o.minThFAST = minTh;
    o.scaleFactor = scaleFactor;
    o.mvScaleFactor.assign(nlevels, 0.f); o.mvLevelSigma2.assign(nlevels, 0.f);
    o.mvScaleFactor[0] = 1.0f; o.mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) {
        o.mvScaleFactor[i] = (float)(o.mvScaleFactor[i - 1] * o.scaleFactor);
        o.mvLevelSigma2[i] = o.mvScaleFactor[i] * o.mvScaleFactor[i];
    }
    o.mvInvScaleFactor.assign(nlevels, 0.f); o.mvInvLevelSigma2.assign(nlevels, 0.f);
    for (int i = 0; i < nlevels; ++i) {
        o.mvInvScaleFactor[i] = 1.0f / o.mvScaleFactor[i];
        o.mvInvLevelSigma2[i] = 1.0f / o.mvLevelSigma2[i];
    }
    o.mnFeaturesPerLevel.assign(nlevels, 0);
    float factor = (float)(1.0f / o.scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; ++level) {
        o.mnFeaturesPerLevel[level] = cv_round(nDesired);
        sumFeatures += o.mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    o.mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);
    // end of each row of the circular patch (S/ORBextractor.cc:463-481)
    o.umax.assign(HALF_PATCH_SIZE + 1, 0);
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) o.umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (o.umax[v0] == o.umax[v0 + 1]) ++v0;
        o.umax[v] = v0;
        ++v0;
    }
}

// ORBextractor::ComputePyramid, S/ORBextractor.cc:1138-1168
void compute_pyramid(Oracle& o, const uint8_t* img, int w, int h, int stride) {
    o.pyramid.resize(o.nlevels);
    for (int level = 0; level < o.nlevels; ++level) {
        float scale = o.mvInvScaleFactor[level];
        Plane& p = o.pyramid[level];
        p.w = cv_round((float)w * scale);
        p.h = cv_round((float)h * scale);
        p.stride = p.w + EDGE_THRESHOLD * 2;
        p.buf.assign((size_t)p.stride * (p.h + EDGE_THRESHOLD * 2), 0);
        if (level != 0) {
            const Plane& q = o.pyramid[level - 1];
            resize_linear_8u(q.roi(), q.w, q.h, q.stride, p.roi(), p.w, p.h, p.stride);
        } else {
            for (int y = 0; y < h; ++y) memcpy(p.roi() + (size_t)y * p.stride, img + (size_t)y * stride, w);
        }
        fill_border(p);
    }
}

// IC_Angle, S/ORBextractor.cc:82-109
float ic_angle(const Plane& p, float ptx, float pty, const std::vector<int>& u_max) {
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = p.roi() + (ptrdiff_t)cv_round(pty) * p.stride + cv_round(ptx);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    int step = p.stride;
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

// computeOrbDescriptor, S/ORBextractor.cc:112-152.  a/b are the correctly rounded single
// precision cos/sin of the float radian angle ("cos/sin definition").
void orb_descriptor(float kx, float ky, float angle_deg, const uint8_t* img, int step,
                    uint8_t* desc) {
    const float factorPI = (float)(M_PI / 180.f);
    float angle = angle_deg * factorPI;
    float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const uint8_t* center = img + (ptrdiff_t)cv_round(ky) * step + cv_round(kx);
    const signed char* pat = slamit_orb_pattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int bit = 0; bit < 8; ++bit) {
            int x0 = pat[bit * 4 + 0], y0 = pat[bit * 4 + 1];
            int x1 = pat[bit * 4 + 2], y1 = pat[bit * 4 + 3];
            int t0 = center[cv_round((double)(x0 * b + y0 * a)) * step + cv_round((double)(x0 * a - y0 * b))];
            int t1 = center[cv_round((double)(x1 * b + y1 * a)) * step + cv_round((double)(x1 * a - y1 * b))];
            val |= (t0 < t1) << bit;
        }
        desc[i] = (uint8_t)val;
    }
}

// ORBextractor::ComputeKeyPointsOctTree, S/ORBextractor.cc:778-873
void compute_keypoints(Oracle& o) {
    o.candidates.assign(o.nlevels, std::vector<int32_t>());
    o.levelKps.assign(o.nlevels, std::vector<slamit_kp>());
    const float W = 30;
    std::vector<FastKp> cell;
    for (int level = 0; level < o.nlevels; ++level) {
        const Plane& p = o.pyramid[level];
        const int minBorderX = EDGE_THRESHOLD - 3;
        const int minBorderY = minBorderX;
        const int maxBorderX = p.w - EDGE_THRESHOLD + 3;
        const int maxBorderY = p.h - EDGE_THRESHOLD + 3;
        std::vector<Key> toDistribute;
        const float width = (float)(maxBorderX - minBorderX);
        const float height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W);
        const int nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) continue;  // the reference divides by zero here
        const int wCell = (int)ceil(width / nCols);
        const int hCell = (int)ceil(height / nRows);
        for (int i = 0; i < nRows; ++i) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; ++j) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                const uint8_t* sub = p.roi() + (ptrdiff_t)(int)iniY * p.stride + (int)iniX;
                int cw = (int)maxX - (int)iniX, ch = (int)maxY - (int)iniY;
                fast9_16(sub, cw, ch, p.stride, o.iniThFAST, cell);
                if (cell.empty()) fast9_16(sub, cw, ch, p.stride, o.minThFAST, cell);
                for (size_t k = 0; k < cell.size(); ++k) {
                    Key kp;
                    kp.x = (float)cell[k].x + j * wCell;
                    kp.y = (float)cell[k].y + i * hCell;
                    kp.response = (float)cell[k].score;
                    toDistribute.push_back(kp);
                    o.candidates[level].push_back((int32_t)kp.x);
                    o.candidates[level].push_back((int32_t)kp.y);
                    o.candidates[level].push_back(cell[k].score);
                }
            }
        }
        std::vector<Key> keys = distribute_octree(toDistribute, minBorderX, maxBorderX, minBorderY,
                                                  maxBorderY, o.mnFeaturesPerLevel[level]);
        const int scaledPatchSize = (int)(PATCH_SIZE * o.mvScaleFactor[level]);
        std::vector<slamit_kp>& out = o.levelKps[level];
        for (size_t i = 0; i < keys.size(); ++i) {
            slamit_kp kp;
            kp.x = keys[i].x + minBorderX;
            kp.y = keys[i].y + minBorderY;
            kp.size = (float)scaledPatchSize;
            kp.angle = -1.f;
            kp.response = keys[i].response;
            kp.octave = level;
            kp.class_id = -1;
            out.push_back(kp);
        }
    }
    for (int level = 0; level < o.nlevels; ++level)
        for (size_t i = 0; i < o.levelKps[level].size(); ++i) {
            slamit_kp& kp = o.levelKps[level][i];
            kp.angle = ic_angle(o.pyramid[level], kp.x, kp.y, o.umax);
        }
}

// ORBextractor::operator(), S/ORBextractor.cc:1064-1136
void run(Oracle& o, const uint8_t* img, int w, int h, int stride) {
    o.kps.clear();
    o.desc.clear();
    o.blurred.assign(o.nlevels, std::vector<uint8_t>());
    if (w <= 0 || h <= 0 || !img) { o.pyramid.clear(); o.candidates.clear(); o.levelKps.clear(); return; }
    compute_pyramid(o, img, w, h, stride);
    compute_keypoints(o);
    for (int level = 0; level < o.nlevels; ++level) {
        std::vector<slamit_kp>& lk = o.levelKps[level];
        if (lk.empty()) continue;
        const Plane& p = o.pyramid[level];
        o.blurred[level].assign((size_t)p.w * p.h, 0);
        gauss7x7_8u(p.roi(), p.w, p.h, p.stride, &o.blurred[level][0], p.w);
        size_t base = o.desc.size();
        o.desc.resize(base + lk.size() * 32);
        for (size_t i = 0; i < lk.size(); ++i)
            orb_descriptor(lk[i].x, lk[i].y, lk[i].angle, &o.blurred[level][0], p.w, &o.desc[base + i * 32]);
        float scale = o.mvScaleFactor[level];
        for (size_t i = 0; i < lk.size(); ++i) {
            slamit_kp kp = lk[i];
            if (level != 0) { kp.x *= scale; kp.y *= scale; }
            o.kps.push_back(kp);
        }
    }
}

// ORBmatcher::DescriptorDistance, S/ORBmatcher.cc:1651-1667 (SWAR popcount over 8 x uint32)
int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

}  // namespace

extern "C" {

void* orb_oracle_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    Oracle* o = new Oracle();
    oracle_init(*o, nfeatures, scaleFactor, nlevels, iniTh, minTh);
    return o;
}
void orb_oracle_destroy(void* h) { delete (Oracle*)h; }

void orb_oracle_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                       int32_t* per_level, int32_t* umax16) {
    Oracle& o = *(Oracle*)h;
    for (int i = 0; i < o.nlevels; ++i) {
        if (scale) scale[i] = o.mvScaleFactor[i];
        if (inv_scale) inv_scale[i] = o.mvInvScaleFactor[i];
        if (sigma2) sigma2[i] = o.mvLevelSigma2[i];
        if (inv_sigma2) inv_sigma2[i] = o.mvInvLevelSigma2[i];
        if (per_level) per_level[i] = o.mnFeaturesPerLevel[i];
    }
    if (umax16) for (int i = 0; i < 16; ++i) umax16[i] = o.umax[i];
}

// Full extraction. Returns the number of keypoints (or -needed if cap is too small).
int orb_oracle_extract(void* h, const uint8_t* img, int w, int hh, int stride, slamit_kp* kps,
                       uint8_t* desc, int cap) {
    Oracle& o = *(Oracle*)h;
    run(o, img, w, hh, stride);
    int n = (int)o.kps.size();
    if (n > cap) return -n;
    if (n) {
        memcpy(kps, &o.kps[0], sizeof(slamit_kp) * n);
        memcpy(desc, &o.desc[0], (size_t)32 * n);
    }
    return n;
}

int orb_oracle_level_size(void* h, int level, int* w, int* hh) {
    Oracle& o = *(Oracle*)h;
    if (level < 0 || level >= (int)o.pyramid.size()) return -1;
    *w = o.pyramid[level].w; *hh = o.pyramid[level].h;
    return 0;
}
// copies the padded plane ((w+38)*(h+38) bytes)
int orb_oracle_level(void* h, int level, uint8_t* dst) {
    Oracle& o = *(Oracle*)h;
    if (level < 0 || level >= (int)o.pyramid.size()) return -1;
    memcpy(dst, &o.pyramid[level].buf[0], o.pyramid[level].buf.size());
    return 0;
}
int orb_oracle_blurred(void* h, int level, uint8_t* dst) {  // w*h bytes; -1 if level had no kps
    Oracle& o = *(Oracle*)h;
    if (level < 0 || level >= (int)o.blurred.size() || o.blurred[level].empty()) return -1;
    memcpy(dst, &o.blurred[level][0], o.blurred[level].size());
    return 0;
}
int orb_oracle_candidates(void* h, int level, int32_t* xys, int cap) {
    Oracle& o = *(Oracle*)h;
    if (level < 0 || level >= (int)o.candidates.size()) return -1;
    int n = (int)o.candidates[level].size() / 3;
    if (xys) memcpy(xys, o.candidates[level].data(), sizeof(int32_t) * 3 * std::min(n, cap));
    return n;
}
int orb_oracle_level_kps(void* h, int level, slamit_kp* kps, int cap) {
    Oracle& o = *(Oracle*)h;
    if (level < 0 || level >= (int)o.levelKps.size()) return -1;
    int n = (int)o.levelKps[level].size();
    if (kps) memcpy(kps, o.levelKps[level].data(), sizeof(slamit_kp) * std::min(n, cap));
    return n;
}

// ---- primitives exposed for known-answer tests ----
void orb_oracle_resize(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                       int dstride) {
    resize_linear_8u(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orb_oracle_blur(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    gauss7x7_8u(src, w, h, sstride, dst, dstride);
}
void orb_oracle_gauss_taps(int32_t* taps7) { int k[7]; gauss7_taps(k); for (int i = 0; i < 7; ++i) taps7[i] = k[i]; }
float orb_oracle_fast_atan2(float y, float x) { return fast_atan2(y, x); }
int orb_oracle_round(double v) { return cv_round(v); }
// cv::FAST on a sub image; returns count, writes x,y,score triplets (raster order)
int orb_oracle_fast(const uint8_t* img, int cols, int rows, int step, int threshold, int32_t* xys, int cap) {
    std::vector<FastKp> out;
    fast9_16(img, cols, rows, step, threshold, out);
    int n = (int)out.size();
    for (int i = 0; i < n && i < cap; ++i) { xys[3 * i] = out[i].x; xys[3 * i + 1] = out[i].y; xys[3 * i + 2] = out[i].score; }
    return n;
}
// DistributeOctTree on (x,y,response) float triplets; returns count, writes triplets in list order
int orb_oracle_octree(const float* xyr, int n, int minX, int maxX, int minY, int maxY, int N, float* out, int cap) {
    std::vector<Key> in(n);
    for (int i = 0; i < n; ++i) { in[i].x = xyr[3 * i]; in[i].y = xyr[3 * i + 1]; in[i].response = xyr[3 * i + 2]; }
    std::vector<Key> res = distribute_octree(in, minX, maxX, minY, maxY, N);
    int m = (int)res.size();
    for (int i = 0; i < m && i < cap; ++i) { out[3 * i] = res[i].x; out[3 * i + 1] = res[i].y; out[3 * i + 2] = res[i].response; }
    return m;
}
void orb_oracle_descriptor(float kx, float ky, float angle_deg, const uint8_t* img, int step, uint8_t* desc) {
    orb_descriptor(kx, ky, angle_deg, img, step, desc);
}
// (float)cos((double)x), (float)sin((double)x) — the oracle's a/b definition
void orb_oracle_cossin(float angle_rad, float* c, float* s) { *c = (float)cos((double)angle_rad); *s = (float)sin((double)angle_rad); }

int orb_oracle_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

// best / second-best with the reference's rule (S/ORBmatcher.cc:1404-1428):
//   if d < best {second = best; best = d; idx = j} else if d < second {second = d}
void orb_oracle_best2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* best_idx,
                      int32_t* best, int32_t* second) {
    for (int i = 0; i < nq; ++i) {
        int bestDist = 256, bestDist2 = 256, bestIdx = -1;
        for (int j = 0; j < nt; ++j) {
            int d = descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx = j; }
            else if (d < bestDist2) bestDist2 = d;
        }
        best_idx[i] = bestIdx; best[i] = bestDist; second[i] = bestDist2;
    }
}
// MapPoint::ComputeDistinctiveDescriptors, S/MapPoint.cc:248-314 (float Distances[N][N], sorted row,
// median = vDists[0.5*(N-1)], strict '<' so the first least-median row wins)
void orb_oracle_distinctive(const uint8_t* desc, const int32_t* offsets, int npoints, int32_t* best_idx, int32_t* best_median) {
    for (int p = 0; p < npoints; ++p) {
        const int o = offsets[p], N = offsets[p + 1] - o;
        if (N <= 0) { best_idx[p] = -1; best_median[p] = 0; continue; }
        std::vector<float> Distances((size_t)N * N);
        for (int i = 0; i < N; ++i) {
            Distances[(size_t)i * N + i] = 0;
            for (int j = i + 1; j < N; ++j) {
                int d = descriptor_distance(desc + 32 * (size_t)(o + i), desc + 32 * (size_t)(o + j));
                Distances[(size_t)i * N + j] = (float)d;
                Distances[(size_t)j * N + i] = (float)d;
            }
        }
        int BestMedian = 2147483647, BestIdx = 0;
        for (int i = 0; i < N; ++i) {
            std::vector<int> vDists(Distances.begin() + (size_t)i * N, Distances.begin() + (size_t)(i + 1) * N);
            std::sort(vDists.begin(), vDists.end());
            int median = vDists[(size_t)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = i; }
        }
        best_idx[p] = BestIdx; best_median[p] = BestMedian;
    }
}
void orb_oracle_matrix(const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* out) {
    for (int i = 0; i < nq; ++i)
        for (int j = 0; j < nt; ++j) out[(size_t)i * nt + j] = (uint16_t)descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
}

}  // extern "C"

// ---- guided search: Frame grid + GetFeaturesInArea + the SearchByProjection loop body ----------
// Frame::AssignFeaturesToGrid (Frame.cc:336-357), Frame::PosInGrid (:505-517),
// Frame::GetFeaturesInArea (:447-502), ORBmatcher::SearchByProjection (ORBmatcher.cc:47-131; the
// frame-to-frame variant :1332-1474 is the same loop with use_ratio = 0).  Mono.
namespace {
const int kGridCols = 64, kGridRows = 48;  // Frame.h:40-41
struct FrameGrid {
    std::vector<int> cell[kGridCols][kGridRows];
};
bool pos_in_grid(float px, float py, float minx, float miny, float invw, float invh, int& posX, int& posY) {
    posX = (int)roundf((px - minx) * invw);
    posY = (int)roundf((py - miny) * invh);
    if (posX < 0 || posX >= kGridCols || posY < 0 || posY >= kGridRows) return false;
    return true;
}
void features_in_area(const FrameGrid& g, const float* kp_xy, const int32_t* kp_octave, float minx, float miny, float invw,
                      float invh, float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) {
    out.clear();
    const int nMinCellX = std::max(0, (int)floorf((x - minx - r) * invw));
    if (nMinCellX >= kGridCols) return;
    const int nMaxCellX = std::min((int)kGridCols - 1, (int)ceilf((x - minx + r) * invw));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)floorf((y - miny - r) * invh));
    if (nMinCellY >= kGridRows) return;
    const int nMaxCellY = std::min((int)kGridRows - 1, (int)ceilf((y - miny + r) * invh));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const std::vector<int>& vCell = g.cell[ix][iy];
            for (size_t j = 0; j < vCell.size(); j++) {
                const int k = vCell[j];
                if (bCheckLevels) {
                    if (kp_octave[k] < minLevel) continue;
                    if (maxLevel >= 0)
                        if (kp_octave[k] > maxLevel) continue;
                }
                const float distx = kp_xy[2 * k] - x, disty = kp_xy[2 * k + 1] - y;
                if (fabsf(distx) < r && fabsf(disty) < r) out.push_back(k);
            }
        }
}
}  // namespace

extern "C" int orb_oracle_guided_search(int n, const float* kp_xy, const int32_t* kp_octave, const uint8_t* kp_desc,
                                        const uint8_t* kp_taken, float minx, float miny, float invw, float invh, int m,
                                        const float* uvr, const int32_t* lmin, const int32_t* lmax, const uint8_t* qdesc,
                                        const uint8_t* valid, const uint8_t* takes, int th_dist, int use_ratio, float nnratio,
                                        float chi2_gate, const float* inv_level_sigma2, int32_t* match_kp, int32_t* out4) {
    FrameGrid* g = new FrameGrid;
    for (int i = 0; i < n; i++) {
        int gx, gy;
        if (pos_in_grid(kp_xy[2 * i], kp_xy[2 * i + 1], minx, miny, invw, invh, gx, gy)) g->cell[gx][gy].push_back(i);
    }
    std::vector<uint8_t> has_mp(kp_taken, kp_taken + n);
    std::vector<int> vIndices;
    int nmatches = 0;
    for (int q = 0; q < m; q++) {
        match_kp[q] = -1;
        out4[4 * q] = 256; out4[4 * q + 1] = -1; out4[4 * q + 2] = 256; out4[4 * q + 3] = -1;
        if (!valid[q]) continue;
        features_in_area(*g, kp_xy, kp_octave, minx, miny, invw, invh, uvr[3 * q], uvr[3 * q + 1], uvr[3 * q + 2], lmin[q], lmax[q], vIndices);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t j = 0; j < vIndices.size(); j++) {
            const int idx = vIndices[j];
            if (has_mp[idx]) continue;
            if (chi2_gate > 0.f) {   // ORBmatcher::Fuse (ORBmatcher.cc:925-936, mono branch): reprojection gate per candidate
                const float ex = uvr[3 * q] - kp_xy[2 * idx], ey = uvr[3 * q + 1] - kp_xy[2 * idx + 1];
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_level_sigma2[kp_octave[idx] & 15] > chi2_gate) continue;
            }
            const int dist = descriptor_distance(qdesc + 32 * (size_t)q, kp_desc + 32 * (size_t)idx);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = kp_octave[idx];
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = kp_octave[idx];
                bestDist2 = dist;
            }
        }
        out4[4 * q] = bestDist; out4[4 * q + 1] = bestLevel; out4[4 * q + 2] = bestDist2; out4[4 * q + 3] = bestLevel2;
        if (bestDist <= th_dist) {
            if (use_ratio && bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            match_kp[q] = bestIdx;
            if (!takes || takes[q]) has_mp[bestIdx] = 1;
            nmatches++;
        }
    }
    delete g;
    return nmatches;
}



// ---- Frame::UndistortKeyPoints / ComputeImageBounds / AssignFeaturesToGrid ---------------------------------------
// cv::undistortPoints(src, dst, K, D, Mat(), K) (called at Frame.cc:548, :572) is OpenCV (2.4.9 here), not in the
// reference tree: restated from the published cvUndistortPoints — normalise, 5 fixed-point iterations of the inverse
// Brown model in double, re-project with P = K (R = I).  PARITY UNPINNED like the other OpenCV primitives.
// Frame::UndistortKeyPoints returns the input untouched when k1 == 0 (Frame.cc:531-535).
extern "C" void orb_oracle_undistort(const float* cam9 /*fx fy cx cy k1 k2 p1 p2 k3*/, const float* xy_in, int n, float* xy_out) {
    const double fx = cam9[0], fy = cam9[1], cx = cam9[2], cy = cam9[3];
    const double k[8] = {cam9[4], cam9[5], cam9[6], cam9[7], cam9[8], 0., 0., 0.};
    const double ifx = 1. / fx, ify = 1. / fy;
    for (int i = 0; i < n; i++) {
        double x = xy_in[2 * i], y = xy_in[2 * i + 1];
        const double x0 = x = (x - cx) * ifx, y0 = y = (y - cy) * ify;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        // RR = P * R = K: xx = fx x + 0 y + cx, ww = 1 / (0 x + 0 y + 1)
        const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);
        xy_out[2 * i] = (float)(xx * ww);
        xy_out[2 * i + 1] = (float)(yy * ww);
    }
}

// Frame::UndistortKeyPoints (Frame.cc:529-559) + Frame::AssignFeaturesToGrid (:336-357): kps_un = kps with the
// undistorted pt; mGrid[x][y] as CSR over cells x * 48 + y, indices in push_back (= keypoint) order.
extern "C" int orb_oracle_frame_finish(const float* cam9, const slamit_kp* kps, int n, float minx, float miny, float invw, float invh,
                                       slamit_kp* kps_un, int32_t* cell_start /*[64*48+1]*/, int32_t* cell_items /*[n]*/) {
    std::vector<float> in(2 * (size_t)std::max(n, 1)), out(2 * (size_t)std::max(n, 1));
    for (int i = 0; i < n; i++) { in[2 * i] = kps[i].x; in[2 * i + 1] = kps[i].y; }
    if (cam9[4] == 0.0f) out = in;
    else orb_oracle_undistort(cam9, in.data(), n, out.data());
    for (int i = 0; i < n; i++) { kps_un[i] = kps[i]; kps_un[i].x = out[2 * i]; kps_un[i].y = out[2 * i + 1]; }
    FrameGrid* g = new FrameGrid;
    for (int i = 0; i < n; i++) {
        int gx, gy;
        if (pos_in_grid(kps_un[i].x, kps_un[i].y, minx, miny, invw, invh, gx, gy)) g->cell[gx][gy].push_back(i);
    }
    int m = 0;
    for (int x = 0; x < kGridCols; x++)
        for (int y = 0; y < kGridRows; y++) {
            cell_start[x * kGridRows + y] = m;
            for (size_t j = 0; j < g->cell[x][y].size(); j++) cell_items[m++] = g->cell[x][y][j];
        }
    cell_start[kGridCols * kGridRows] = m;
    delete g;
    return m;
}

// ORBmatcher::SearchForInitialization (ORBmatcher.cc:409-474): the matching loop, without the rotation histogram
// (:476-518, host logic in the shim).  F1 keypoints above level 0 are skipped; window = windowSize at level 0 of F2;
// vMatchedDistance gates candidates; an accepted match takes the keypoint over from an earlier query.
extern "C" int orb_oracle_search_init(int n1, const int32_t* kp1_octave, const uint8_t* desc1, const float* prev_xy, int n2,
                                      const float* kp2_xy, const int32_t* kp2_octave, const uint8_t* desc2, float minx, float miny,
                                      float invw, float invh, int windowSize, float nnratio, int th_low, int32_t* vnMatches12,
                                      int32_t* accepted /* bestIdx2 at the query's own turn, -1 if none (what :467-477 bins) */) {
    FrameGrid* g = new FrameGrid;
    for (int i = 0; i < n2; i++) {
        int gx, gy;
        if (pos_in_grid(kp2_xy[2 * i], kp2_xy[2 * i + 1], minx, miny, invw, invh, gx, gy)) g->cell[gx][gy].push_back(i);
    }
    int nmatches = 0;
    for (int i = 0; i < n1; i++) { vnMatches12[i] = -1; accepted[i] = -1; }
    std::vector<int> vMatchedDistance(n2, INT_MAX), vnMatches21(n2, -1);
    std::vector<int> vIndices2;
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = kp1_octave[i1];
        if (level1 > 0) continue;
        features_in_area(*g, kp2_xy, kp2_octave, minx, miny, invw, invh, prev_xy[2 * i1], prev_xy[2 * i1 + 1], (float)windowSize, level1, level1, vIndices2);
        if (vIndices2.empty()) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (size_t j = 0; j < vIndices2.size(); j++) {
            const int i2 = vIndices2[j];
            const int dist = descriptor_distance(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= th_low) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                accepted[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
            }
        }
    }
    delete g;
    return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------
// The BoW drivers' matching loops, over node groups that the caller has already intersected (the while / lower_bound
// walk over the two DBoW2::FeatureVectors, ORBmatcher.cc:184-278, is host logic in the shim).  Sequential like the
// reference: groups in order, queries in order, candidates in order.
//   mode 0: ORBmatcher::SearchByBoW(KeyFrame*, Frame&) :187-262 (th_inclusive = 1, valid2 = NULL) and
//           ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*) :558-628 (th_inclusive = 0, valid2 = good MapPoint)
//   mode 1: ORBmatcher::SearchForTriangulation :697-782 (monocular: bStereo1 = bStereo2 = false, bOnlyStereo = false)
//           with CheckDistEpipolarLine :141-158.  vbMatched2 is declared by the reference and never set: no exclusion.
// The rotation histogram that follows (:264-287 / :630-654 / :784-808) is host logic in the shim.  PARITY UNPINNED
// for the float gates of mode 1 (whether the reference's compiler contracts a*b+c is not known; none here).
extern "C" int orb_oracle_bow_search(int mode, const uint8_t* desc1, int n1, const uint8_t* valid1, const uint8_t* desc2, int n2,
                                     const uint8_t* valid2, int n_groups, const int32_t* q_ptr, const int32_t* q_idx,
                                     const int32_t* c_ptr, const int32_t* c_idx, int th, int th_inclusive, float nnratio,
                                     const float* F12, float ex, float ey, const float* kp1_xy, const float* kp2_xy,
                                     const int32_t* kp2_octave, const float* scale_factor, const float* level_sigma2,
                                     int32_t* match12, int32_t* dist12) {
    int nmatches = 0;
    for (int i = 0; i < n1; i++) { match12[i] = -1; dist12[i] = 256; }
    std::vector<bool> vbMatched2((size_t)n2, false);
    for (int g = 0; g < n_groups; g++) {
        for (int a = q_ptr[g]; a < q_ptr[g + 1]; a++) {
            const int idx1 = q_idx[a];
            if (valid1 && !valid1[idx1]) continue;
            const uint8_t* d1 = desc1 + 32 * (size_t)idx1;
            if (mode == 0) {
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int b = c_ptr[g]; b < c_ptr[g + 1]; b++) {
                    const int idx2 = c_idx[b];
                    if (vbMatched2[idx2] || (valid2 && !valid2[idx2])) continue;
                    const int dist = descriptor_distance(d1, desc2 + 32 * (size_t)idx2);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestIdx2 >= 0) dist12[idx1] = bestDist1;
                // (bestIdx2 < 0 can only get here with th >= 256, outside the reference's TH_LOW = 50; the reference would
                //  index its match vector with -1 there)
                if (bestIdx2 >= 0 && (th_inclusive ? bestDist1 <= th : bestDist1 < th)) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        match12[idx1] = bestIdx2;
                        vbMatched2[bestIdx2] = true;
                        nmatches++;
                    }
                }
            } else {
                const float x1 = kp1_xy[2 * idx1], y1 = kp1_xy[2 * idx1 + 1];
                int bestDist = th, bestIdx2 = -1;
                for (int b = c_ptr[g]; b < c_ptr[g + 1]; b++) {
                    const int idx2 = c_idx[b];
                    if (valid2 && !valid2[idx2]) continue;
                    const int dist = descriptor_distance(d1, desc2 + 32 * (size_t)idx2);
                    if (dist > th || dist > bestDist) continue;
                    const float x2 = kp2_xy[2 * idx2], y2 = kp2_xy[2 * idx2 + 1];
                    const int oct = kp2_octave[idx2];
                    const float distex = ex - x2;
                    const float distey = ey - y2;
                    if (distex * distex + distey * distey < 100 * scale_factor[oct]) continue;
                    // CheckDistEpipolarLine
                    const float la = x1 * F12[0] + y1 * F12[3] + F12[6];
                    const float lb = x1 * F12[1] + y1 * F12[4] + F12[7];
                    const float lc = x1 * F12[2] + y1 * F12[5] + F12[8];
                    const float num = la * x2 + lb * y2 + lc;
                    const float den = la * la + lb * lb;
                    if (den == 0) continue;
                    const float dsqr = num * num / den;
                    if (dsqr < 3.84 * level_sigma2[oct]) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) { match12[idx1] = bestIdx2; dist12[idx1] = bestDist; nmatches++; }
            }
        }
    }
    return nmatches;
}
