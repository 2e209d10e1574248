#include "computation_model.h"

namespace BaSpaCho {

// Constants = data published in the reference (ComputationModel.cpp:12-31).
const ComputationModel ComputationModel::model_OpenBlas_i7_1185g7{
    {3.527141723946874224e-07, -5.382557351808083451e-08, 4.677984682984275924e-09,
     7.384424667338682676e-12},
    {1.101115592925888909e-06, 6.936563076265144074e-07, -1.827661167503034051e-09,
     1.959826916788009885e-09, 1.079857543323972179e-09, 2.963338652996178598e-11},
    {6.14190596709488416e-07, -4.489948374364910256e-09, 5.943145978912038475e-10,
     -1.201283634136652872e-08, 1.266858215451465993e-09, 2.624001993284897048e-11},
    {3.069607518266660019e-07, 3.763778311956422235e-08, 1.991443920635728855e-07,
     3.788938150548870089e-09}};

const ComputationModel ComputationModel::model_Cuda117_2080Ti{
    {1.826148710881364949e-05, 4.44453427795126131e-06, 4.824038596816424795e-09,
     2.616891884279722513e-13},
    {2.252999933361577123e-06, -3.090170544094683306e-08, 5.719994050518756252e-10,
     -1.007960791939421421e-09, 1.167012989682758675e-10, -4.025574444690399163e-13},
    {4.55174422740625352e-06, 8.155647192215145375e-11, -1.76305533240800307e-14,
     1.054095249414374123e-08, -2.491863357081539836e-12, -1.095648600879501635e-16},
    {1.975089750288875748e-06, -1.339369810950508464e-10, -3.758728373628488434e-10,
     1.745285595679570848e-13}};

// MI355X / level-scheduled HIP backend.  Round 6: re-fitted on the round-6 kernels -- the least-squares
// fit of the per-op samples taken on an MI355X (tools/op_stats_dump.py -> profiles/r06_opstats_*.csv ->
// tools/fit_computation_model.py -> profiles/r06_model_fit.json, the pipeline of Bench.cpp:72-124 +
// examples/OptimizeCompModel.cpp:64-275) with every op's CONSTANT term multiplied by the level-batching
// share 0.03.  What the share means, and why it is not zero: the samples time an op as a launch of
// its own, while a level of the fused path runs the ops of all its lumps (and of all matrices of a
// batch) in ONE launch, so the marginal fixed cost of one more lump is a few per cent of a kernel
// boundary.  profiles/r06_model_batch_sweep.txt is the measurement: plans from "no merge that adds
// fill" (share 0: 2.45 GF, 42 levels) to "everything the per-op driver would merge" (share 1: 6.5 GF,
// 51 levels) on GRID 82x82 for batches of 1 / 8 / 64 -- the optimum is flat between shares 0.01 and
// 0.03 for ALL three batch sizes (within 1 %), an explicit level term (csrc/elimination_tree.cpp) and
// the expected batch size (HipBackendOptions::expectedBatch) change the partition by 0-2 lumps of
// 1446, and no plan within +-20 % of the flops is faster: the level count of such a structure is its
// elimination tree's height in columns / 64, which merging cannot shorten.
const ComputationModel ComputationModel::model_Hip_MI355X{
    // potrf: a + b n + c n^2 + d n^3
    {4.842342370685720e-08, 2.543247713289900e-07, 6.834653829137758e-11, 9.975328171884997e-16},
    // trsm: a + b n + c n^2 + (d + e n + f n^2) k
    {1.645234651723887e-07, 1.095388935580508e-07, 3.684825255567626e-10, 7.028454342479416e-10, 0.000000000000000e+00, 0.000000000000000e+00},
    // syge: a + b u + c v + k (d + e u + f v)
    {2.091125895395996e-07, 5.866009986064752e-10, 1.749721373224342e-12, 5.301820950214163e-08, 0.000000000000000e+00, 8.030970513286191e-15},
    // asmbl: a + b br + c bc + d br bc
    {1.768453072613642e-07, 0.000000000000000e+00, 1.884331035327229e-08, 1.103342015203656e-09}};

}  // namespace BaSpaCho
