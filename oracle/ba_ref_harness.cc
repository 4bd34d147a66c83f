// ba_ref_harness.cc — drives the REFERENCE's own g2o (compiled by oracle/Makefile.ref from
// /root/reference, outputs in oracle/_ref/) exactly the way Optimizer::LocalBundleAdjustment
// does, but from POD inputs.  TEST INFRASTRUCTURE ONLY: used in the authoring container to
// validate oracle/ba_oracle.cc and to generate tests/golden/ba_*.npz (tools/gen_ba_golden.py).
//
// Follows ORB_SLAM2/src/Optimizer.cc ("S/Optimizer.cc"):
//   solver stack         :507-515   BlockSolver_6_3 + LinearSolverEigen + Levenberg
//   keyframe vertices    :523-546   id = index, fixed flag, estimate = Converter::toSE3Quat(pose)
//   point vertices       :572-580   id = index + maxKFid + 1, marginalised
//   mono edges           :596-619   obs, Omega = I*invSigma2, Huber(delta), fx fy cx cy
//   schedule             :659-707   optimize(5) robust -> chi2 gate / depth -> level 1, kernels off
//                                   -> initializeOptimization(0) -> optimize(10)
//   erasure test         :715-728   chi2 > gate || !isDepthPositive on EVERY edge
#include <vector>

#include "Thirdparty/g2o/g2o/core/block_solver.h"
#include "Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.h"
#include "Thirdparty/g2o/g2o/core/robust_kernel_impl.h"
#include "Thirdparty/g2o/g2o/core/sparse_optimizer.h"
#include "Thirdparty/g2o/g2o/core/hyper_graph_action.h"
#include "Thirdparty/g2o/g2o/solvers/linear_solver_dense.h"
#include "Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h"
#include "Thirdparty/g2o/g2o/types/types_six_dof_expmap.h"
#include "Thirdparty/g2o/g2o/types/types_seven_dof_expmap.h"

#include "../include/slamit.h"

namespace {

// Post-iteration action: records lambda, the number of LM trials and the robust cost of the
// LAST EVALUATED trial (no recomputation: batch statistics stay off like in the reference, so
// the per-edge errors keep the values the reference's chi2 gate reads at S/Optimizer.cc:680).
struct LambdaRecorder : public g2o::HyperGraphAction {
    g2o::OptimizationAlgorithmLevenberg* alg;
    g2o::SparseOptimizer* opt;
    std::vector<double> lambdas, chi2;
    std::vector<int> trials;
    virtual g2o::HyperGraphAction* operator()(const g2o::HyperGraph*, Parameters* = 0) {
        lambdas.push_back(alg->currentLambda());
        trials.push_back(alg->levenbergIteration());
        chi2.push_back(opt->activeRobustChi2());
        return this;
    }
};

}  // namespace

extern "C" int ba_ref_solve(const slamit_ba_problem* pb, const slamit_ba_opts* op, slamit_ba_result* res) {
    g2o::SparseOptimizer optimizer;
    g2o::BlockSolver_6_3::LinearSolverType* linearSolver =
        new g2o::LinearSolverEigen<g2o::BlockSolver_6_3::PoseMatrixType>();
    g2o::BlockSolver_6_3* solver_ptr = new g2o::BlockSolver_6_3(linearSolver);
    g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(solver_ptr);
    optimizer.setAlgorithm(solver);
    bool stopflag = false;
    if (op->stop) optimizer.setForceStopFlag(const_cast<bool*>(reinterpret_cast<const volatile bool*>(op->stop)));
    (void)stopflag;

    const int K = pb->n_kf, P = pb->n_pt, E = pb->n_edge;
    for (int k = 0; k < K; ++k) {
        Eigen::Matrix<double, 3, 3> R;
        const double* p = pb->kf_pose + 12 * k;
        R << p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8];
        Eigen::Matrix<double, 3, 1> t(p[9], p[10], p[11]);
        g2o::VertexSE3Expmap* v = new g2o::VertexSE3Expmap();
        v->setEstimate(g2o::SE3Quat(R, t));
        v->setId(k);
        v->setFixed(pb->kf_fixed[k] != 0);
        optimizer.addVertex(v);
    }
    const int maxKFid = K - 1;
    std::vector<g2o::EdgeSE3ProjectXYZ*> edges(E, (g2o::EdgeSE3ProjectXYZ*)0);
    std::vector<g2o::EdgeStereoSE3ProjectXYZ*> sedges(E, (g2o::EdgeStereoSE3ProjectXYZ*)0);   // stereo observations (S/Optimizer.cc:621-650)
    const double delta_s = op->huber_delta_stereo > 0 ? op->huber_delta_stereo : (double)(float)sqrt(7.815);
    const double gate_s = op->chi2_gate_stereo > 0 ? op->chi2_gate_stereo : 7.815;
    for (int p = 0; p < P; ++p) {
        g2o::VertexSBAPointXYZ* v = new g2o::VertexSBAPointXYZ();
        v->setEstimate(Eigen::Vector3d(pb->pt_xyz[3 * p], pb->pt_xyz[3 * p + 1], pb->pt_xyz[3 * p + 2]));
        v->setId(p + maxKFid + 1);
        v->setMarginalized(true);
        optimizer.addVertex(v);
    }
    for (int e = 0; e < E; ++e) {
        const int id = pb->edge_pt[e] + maxKFid + 1, kf = pb->edge_kf[e];
        if (pb->edge_ur && !(pb->edge_ur[e] < 0)) {
            Eigen::Matrix<double, 3, 1> obs;
            obs << pb->edge_uv[2 * e], pb->edge_uv[2 * e + 1], pb->edge_ur[e];
            g2o::EdgeStereoSE3ProjectXYZ* ed = new g2o::EdgeStereoSE3ProjectXYZ();
            ed->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(id)));
            ed->setVertex(1, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(kf)));
            ed->setMeasurement(obs);
            Eigen::Matrix3d Info = Eigen::Matrix3d::Identity() * pb->edge_inv_sigma2[e];
            ed->setInformation(Info);
            g2o::RobustKernelHuber* rk = new g2o::RobustKernelHuber;
            ed->setRobustKernel(rk);
            rk->setDelta(delta_s);
            ed->fx = pb->kf_intr[4 * kf]; ed->fy = pb->kf_intr[4 * kf + 1];
            ed->cx = pb->kf_intr[4 * kf + 2]; ed->cy = pb->kf_intr[4 * kf + 3];
            ed->bf = pb->kf_bf[kf];
            optimizer.addEdge(ed);
            sedges[e] = ed;
            continue;
        }
        Eigen::Matrix<double, 2, 1> obs;
        obs << pb->edge_uv[2 * e], pb->edge_uv[2 * e + 1];
        g2o::EdgeSE3ProjectXYZ* ed = new g2o::EdgeSE3ProjectXYZ();
        ed->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(id)));
        ed->setVertex(1, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(kf)));
        ed->setMeasurement(obs);
        ed->setInformation(Eigen::Matrix2d::Identity() * pb->edge_inv_sigma2[e]);
        g2o::RobustKernelHuber* rk = new g2o::RobustKernelHuber;
        ed->setRobustKernel(rk);
        rk->setDelta(op->huber_delta);
        ed->fx = pb->kf_intr[4 * kf]; ed->fy = pb->kf_intr[4 * kf + 1];
        ed->cx = pb->kf_intr[4 * kf + 2]; ed->cy = pb->kf_intr[4 * kf + 3];
        optimizer.addEdge(ed);
        edges[e] = ed;
    }

    slamit_ba_stats* st = res->stats;
    if (st) memset(st, 0, sizeof(*st));
    LambdaRecorder rec;
    rec.alg = solver;
    rec.opt = &optimizer;
    optimizer.addPostIterationAction(&rec);

    for (int stage = 0; stage < 2; ++stage) {
        const int its = stage == 0 ? op->its_robust : op->its_final;
        if (op->stop && *op->stop) break;  // S/Optimizer.cc:655-657, 664-666
        if (stage == 1) {
            for (int e = 0; e < E; ++e) {
                if (sedges[e]) {   // S/Optimizer.cc:689-703
                    g2o::EdgeStereoSE3ProjectXYZ* ed = sedges[e];
                    bool out = ed->chi2() > gate_s || !ed->isDepthPositive();
                    if (out) ed->setLevel(1);
                    if (res->edge_stage1_outlier) res->edge_stage1_outlier[e] = out;
                    ed->setRobustKernel(0);
                    continue;
                }
                g2o::EdgeSE3ProjectXYZ* ed = edges[e];
                bool out = ed->chi2() > op->chi2_gate || !ed->isDepthPositive();
                if (out) ed->setLevel(1);
                if (res->edge_stage1_outlier) res->edge_stage1_outlier[e] = out;
                ed->setRobustKernel(0);
            }
            optimizer.initializeOptimization(0);
        } else {
            optimizer.initializeOptimization();
        }
        rec.lambdas.clear(); rec.trials.clear(); rec.chi2.clear();
        if (st) {
            optimizer.computeActiveErrors();  // harmless: solve() recomputes them first thing
            st->chi2_init[stage] = optimizer.activeRobustChi2();
        }
        int n = optimizer.optimize(its);
        if (n < 0) n = 0;  // "0 vertices to optimize": every edge was de-activated
        if (st) {
            st->n_its[stage] = n;
            for (int i = 0; i < n && i < SLAMIT_BA_MAX_ITS; ++i) {
                st->chi2[stage][i] = i < (int)rec.chi2.size() ? rec.chi2[i] : 0;
                st->lambda[stage][i] = i < (int)rec.lambdas.size() ? rec.lambdas[i] : 0;
                st->trials[stage][i] = i < (int)rec.trials.size() ? rec.trials[i] : 0;
            }
        }
    }

    for (int e = 0; e < E; ++e) {
        if (sedges[e]) {   // S/Optimizer.cc:731-745
            g2o::EdgeStereoSE3ProjectXYZ* ed = sedges[e];
            if (res->edge_chi2) res->edge_chi2[e] = ed->chi2();
            if (res->edge_outlier) res->edge_outlier[e] = ed->chi2() > gate_s || !ed->isDepthPositive();
            continue;
        }
        g2o::EdgeSE3ProjectXYZ* ed = edges[e];
        if (res->edge_chi2) res->edge_chi2[e] = ed->chi2();
        if (res->edge_outlier) res->edge_outlier[e] = ed->chi2() > op->chi2_gate || !ed->isDepthPositive();
    }
    for (int k = 0; k < K; ++k) {
        g2o::VertexSE3Expmap* v = static_cast<g2o::VertexSE3Expmap*>(optimizer.vertex(k));
        Eigen::Matrix<double, 4, 4> T = v->estimate().to_homogeneous_matrix();  // Converter::toCvMat(SE3Quat)
        double* o = res->kf_pose + 12 * k;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) o[3 * r + c] = T(r, c);
        for (int r = 0; r < 3; ++r) o[9 + r] = T(r, 3);
    }
    for (int p = 0; p < P; ++p) {
        g2o::VertexSBAPointXYZ* v = static_cast<g2o::VertexSBAPointXYZ*>(optimizer.vertex(p + maxKFid + 1));
        for (int c = 0; c < 3; ++c) res->pt_xyz[3 * p + c] = v->estimate()[c];
    }
    return 0;
}


// Optimizer::PoseOptimization (S/Optimizer.cc:239-451) driven from POD inputs with the reference's g2o.
extern "C" int pose_ref_solve(const slamit_pose_problem* pb, slamit_pose_result* res) {
    g2o::SparseOptimizer optimizer;
    g2o::BlockSolver_6_3::LinearSolverType* linearSolver = new g2o::LinearSolverDense<g2o::BlockSolver_6_3::PoseMatrixType>();
    g2o::BlockSolver_6_3* solver_ptr = new g2o::BlockSolver_6_3(linearSolver);
    g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(solver_ptr);
    optimizer.setAlgorithm(solver);
    const int N = pb->n;
    Eigen::Matrix<double, 3, 3> R;
    R << pb->pose[0], pb->pose[1], pb->pose[2], pb->pose[3], pb->pose[4], pb->pose[5], pb->pose[6], pb->pose[7], pb->pose[8];
    Eigen::Matrix<double, 3, 1> t(pb->pose[9], pb->pose[10], pb->pose[11]);
    const g2o::SE3Quat Tcw(R, t);
    g2o::VertexSE3Expmap* vSE3 = new g2o::VertexSE3Expmap();
    vSE3->setEstimate(Tcw);
    vSE3->setId(0);
    vSE3->setFixed(false);
    optimizer.addVertex(vSE3);
    for (int r = 0; r < 4; ++r) { res->n_its[r] = 0; res->chi2[r] = 0; }
    std::vector<g2o::EdgeSE3ProjectXYZOnlyPose*> edges;
    std::vector<g2o::EdgeStereoSE3ProjectXYZOnlyPose*> sedges;   // S/Optimizer.cc:319-356
    std::vector<int> index, sindex;                               // vnIndexEdgeMono / vnIndexEdgeStereo
    const float deltaMono = sqrt(5.991);
    const float deltaStereo = sqrt(7.815);
    int nInitialCorrespondences = 0;
    for (int i = 0; i < N; ++i) {
        nInitialCorrespondences++;
        res->outlier[i] = 0;
        if (pb->ur && !(pb->ur[i] < 0)) {
            Eigen::Matrix<double, 3, 1> obs;
            const float kp_ur = (float)pb->ur[i];
            obs << pb->uv[2 * i], pb->uv[2 * i + 1], kp_ur;
            g2o::EdgeStereoSE3ProjectXYZOnlyPose* e = new g2o::EdgeStereoSE3ProjectXYZOnlyPose();
            e->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(0)));
            e->setMeasurement(obs);
            Eigen::Matrix3d Info = Eigen::Matrix3d::Identity() * pb->inv_sigma2[i];
            e->setInformation(Info);
            g2o::RobustKernelHuber* rk = new g2o::RobustKernelHuber;
            e->setRobustKernel(rk);
            rk->setDelta(deltaStereo);
            e->fx = pb->intr[0]; e->fy = pb->intr[1]; e->cx = pb->intr[2]; e->cy = pb->intr[3];
            e->bf = pb->bf;
            e->Xw[0] = pb->xw[3 * i]; e->Xw[1] = pb->xw[3 * i + 1]; e->Xw[2] = pb->xw[3 * i + 2];
            optimizer.addEdge(e);
            sedges.push_back(e);
            sindex.push_back(i);
            continue;
        }
        index.push_back(i);
        Eigen::Matrix<double, 2, 1> obs;
        obs << pb->uv[2 * i], pb->uv[2 * i + 1];
        g2o::EdgeSE3ProjectXYZOnlyPose* e = new g2o::EdgeSE3ProjectXYZOnlyPose();
        e->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(0)));
        e->setMeasurement(obs);
        e->setInformation(Eigen::Matrix2d::Identity() * pb->inv_sigma2[i]);
        g2o::RobustKernelHuber* rk = new g2o::RobustKernelHuber;
        e->setRobustKernel(rk);
        rk->setDelta(deltaMono);
        e->fx = pb->intr[0]; e->fy = pb->intr[1]; e->cx = pb->intr[2]; e->cy = pb->intr[3];
        e->Xw[0] = pb->xw[3 * i]; e->Xw[1] = pb->xw[3 * i + 1]; e->Xw[2] = pb->xw[3 * i + 2];
        optimizer.addEdge(e);
        edges.push_back(e);
    }
    if (nInitialCorrespondences < 3) {
        memcpy(res->pose, pb->pose, sizeof(double) * 12);
        res->n_inliers = 0;
        return 0;
    }
    const float chi2Mono[4] = {5.991, 5.991, 5.991, 5.991};
    const float chi2Stereo[4] = {7.815, 7.815, 7.815, 7.815};
    const int its[4] = {10, 10, 10, 10};
    int nBad = 0;
    for (size_t it = 0; it < 4; it++) {
        vSE3->setEstimate(Tcw);
        optimizer.initializeOptimization(0);
        int n = optimizer.optimize(its[it]);
        res->n_its[it] = n < 0 ? 0 : n;
        res->chi2[it] = n > 0 ? optimizer.activeRobustChi2() : 0.0;
        nBad = 0;
        for (size_t i = 0; i < edges.size(); i++) {
            g2o::EdgeSE3ProjectXYZOnlyPose* e = edges[i];
            const int idx = index[i];
            if (res->outlier[idx]) e->computeError();
            const float chi2 = e->chi2();
            if (chi2 > chi2Mono[it]) { res->outlier[idx] = 1; e->setLevel(1); nBad++; }
            else { res->outlier[idx] = 0; e->setLevel(0); }
            if (it == 2) e->setRobustKernel(0);
        }
        for (size_t i = 0; i < sedges.size(); i++) {
            g2o::EdgeStereoSE3ProjectXYZOnlyPose* e = sedges[i];
            const int idx = sindex[i];
            if (res->outlier[idx]) e->computeError();
            const float chi2 = e->chi2();
            if (chi2 > chi2Stereo[it]) { res->outlier[idx] = 1; e->setLevel(1); nBad++; }
            else { e->setLevel(0); res->outlier[idx] = 0; }
            if (it == 2) e->setRobustKernel(0);
        }
        if (optimizer.edges().size() < 10) break;
    }
    Eigen::Matrix<double, 4, 4> T = static_cast<g2o::VertexSE3Expmap*>(optimizer.vertex(0))->estimate().to_homogeneous_matrix();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) res->pose[3 * r + c] = T(r, c); res->pose[9 + r] = T(r, 3); }
    res->n_inliers = nInitialCorrespondences - nBad;
    return 0;
}


// Optimizer::OptimizeSim3 (S/Optimizer.cc:1046-1247) driven from POD inputs with the reference's g2o: the graph the
// reference builds after its validity tests (:1099-1178), the two-stage schedule and the inlier tests (:1180-1247).
extern "C" int sim3_ref_solve(const slamit_sim3_problem* pb, slamit_sim3_result* res) {
    g2o::SparseOptimizer optimizer;
    g2o::BlockSolverX::LinearSolverType* linearSolver = new g2o::LinearSolverDense<g2o::BlockSolverX::PoseMatrixType>();
    g2o::BlockSolverX* solver_ptr = new g2o::BlockSolverX(linearSolver);
    g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(solver_ptr);
    optimizer.setAlgorithm(solver);
    Eigen::Matrix3d R;
    R << pb->r12[0], pb->r12[1], pb->r12[2], pb->r12[3], pb->r12[4], pb->r12[5], pb->r12[6], pb->r12[7], pb->r12[8];
    const g2o::Sim3 g2oS12(R, Eigen::Vector3d(pb->t12[0], pb->t12[1], pb->t12[2]), pb->s12);
    g2o::VertexSim3Expmap* vSim3 = new g2o::VertexSim3Expmap();
    vSim3->_fix_scale = pb->fix_scale != 0;
    vSim3->setEstimate(g2oS12);
    vSim3->setId(0);
    vSim3->setFixed(false);
    vSim3->_principle_point1[0] = pb->intr1[2]; vSim3->_principle_point1[1] = pb->intr1[3];
    vSim3->_focal_length1[0] = pb->intr1[0]; vSim3->_focal_length1[1] = pb->intr1[1];
    vSim3->_principle_point2[0] = pb->intr2[2]; vSim3->_principle_point2[1] = pb->intr2[3];
    vSim3->_focal_length2[0] = pb->intr2[0]; vSim3->_focal_length2[1] = pb->intr2[1];
    optimizer.addVertex(vSim3);
    const int N = pb->n;
    std::vector<g2o::EdgeSim3ProjectXYZ*> vpEdges12;
    std::vector<g2o::EdgeInverseSim3ProjectXYZ*> vpEdges21;
    const float th2 = (float)pb->th2;
    const float deltaHuber = sqrt(th2);
    int nCorrespondences = 0;
    for (int i = 0; i < N; i++) {
        res->inlier[i] = 1;
        const int id1 = 2 * i + 1, id2 = 2 * (i + 1);
        g2o::VertexSBAPointXYZ* vPoint1 = new g2o::VertexSBAPointXYZ();
        vPoint1->setEstimate(Eigen::Vector3d(pb->p1[3 * i], pb->p1[3 * i + 1], pb->p1[3 * i + 2]));
        vPoint1->setId(id1); vPoint1->setFixed(true);
        optimizer.addVertex(vPoint1);
        g2o::VertexSBAPointXYZ* vPoint2 = new g2o::VertexSBAPointXYZ();
        vPoint2->setEstimate(Eigen::Vector3d(pb->p2[3 * i], pb->p2[3 * i + 1], pb->p2[3 * i + 2]));
        vPoint2->setId(id2); vPoint2->setFixed(true);
        optimizer.addVertex(vPoint2);
        nCorrespondences++;
        Eigen::Matrix<double, 2, 1> obs1;
        obs1 << pb->obs1[2 * i], pb->obs1[2 * i + 1];
        g2o::EdgeSim3ProjectXYZ* e12 = new g2o::EdgeSim3ProjectXYZ();
        e12->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(id2)));
        e12->setVertex(1, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(0)));
        e12->setMeasurement(obs1);
        e12->setInformation(Eigen::Matrix2d::Identity() * pb->inv_sigma2_1[i]);
        g2o::RobustKernelHuber* rk1 = new g2o::RobustKernelHuber;
        e12->setRobustKernel(rk1);
        rk1->setDelta(deltaHuber);
        optimizer.addEdge(e12);
        Eigen::Matrix<double, 2, 1> obs2;
        obs2 << pb->obs2[2 * i], pb->obs2[2 * i + 1];
        g2o::EdgeInverseSim3ProjectXYZ* e21 = new g2o::EdgeInverseSim3ProjectXYZ();
        e21->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(id1)));
        e21->setVertex(1, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(0)));
        e21->setMeasurement(obs2);
        e21->setInformation(Eigen::Matrix2d::Identity() * pb->inv_sigma2_2[i]);
        g2o::RobustKernelHuber* rk2 = new g2o::RobustKernelHuber;
        e21->setRobustKernel(rk2);
        rk2->setDelta(deltaHuber);
        optimizer.addEdge(e21);
        vpEdges12.push_back(e12);
        vpEdges21.push_back(e21);
    }
    for (int k = 0; k < 2; ++k) { res->n_its[k] = 0; res->chi2[k] = 0; }
    memcpy(res->r12, pb->r12, sizeof(res->r12)); memcpy(res->t12, pb->t12, sizeof(res->t12)); res->s12 = pb->s12;
    optimizer.initializeOptimization();
    int it = optimizer.optimize(5);
    res->n_its[0] = it < 0 ? 0 : it;
    res->chi2[0] = it > 0 ? optimizer.activeRobustChi2() : 0.0;
    int nBad = 0;
    for (size_t i = 0; i < vpEdges12.size(); i++) {
        g2o::EdgeSim3ProjectXYZ* e12 = vpEdges12[i];
        g2o::EdgeInverseSim3ProjectXYZ* e21 = vpEdges21[i];
        if (!e12 || !e21) continue;
        if (e12->chi2() > th2 || e21->chi2() > th2) {
            res->inlier[i] = 0;
            optimizer.removeEdge(e12);
            optimizer.removeEdge(e21);
            vpEdges12[i] = static_cast<g2o::EdgeSim3ProjectXYZ*>(NULL);
            vpEdges21[i] = static_cast<g2o::EdgeInverseSim3ProjectXYZ*>(NULL);
            nBad++;
        }
    }
    const int nMoreIterations = nBad > 0 ? 10 : 5;
    if (nCorrespondences - nBad < 10) { res->n_inliers = 0; return 0; }
    optimizer.initializeOptimization();
    it = optimizer.optimize(nMoreIterations);
    res->n_its[1] = it < 0 ? 0 : it;
    res->chi2[1] = it > 0 ? optimizer.activeRobustChi2() : 0.0;
    int nIn = 0;
    for (size_t i = 0; i < vpEdges12.size(); i++) {
        g2o::EdgeSim3ProjectXYZ* e12 = vpEdges12[i];
        g2o::EdgeInverseSim3ProjectXYZ* e21 = vpEdges21[i];
        if (!e12 || !e21) continue;
        if (e12->chi2() > th2 || e21->chi2() > th2) res->inlier[i] = 0;
        else nIn++;
    }
    const g2o::Sim3 out = static_cast<g2o::VertexSim3Expmap*>(optimizer.vertex(0))->estimate();
    const Eigen::Matrix3d Ro = out.rotation().toRotationMatrix();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) res->r12[3 * r + c] = Ro(r, c); res->t12[r] = out.translation()[r]; }
    res->s12 = out.scale();
    res->n_inliers = nIn;
    return 0;
}
