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

// MI355X / level-scheduled HIP backend: the least-squares fit of the per-op samples taken on an
// MI355X (tools/op_stats_dump.py -> profiles/r02_opstats_*.csv -> tools/fit_computation_model.py ->
// profiles/r02_model_fit.json, the pipeline of Bench.cpp:72-124 + examples/OptimizeCompModel.cpp:64-275)
// with every op's CONSTANT term multiplied by kLevelBatchingShare = 0.03: the samples time an op as a
// launch of its own, while a level of the fused path batches the ops of all its lumps into one
// launch, so the marginal fixed cost of one more lump is a small share of a kernel boundary and
// the fill a merge adds is paid in full.  The share was chosen on the device (tools/model_eval.py,
// profiles/r02_model_eval.txt): 64 x GRID 82x82 12.44 ms at the round-1 hand-set constants, 14.6 /
// 11.84 / 11.42 / 11.37 at shares 0.3 / 0.1 / 0.03 / 0.01; FLAT-50k 28.5 -> 27.1; BAL-871 unchanged (its
// camera block is merged by the dense-merge rule of elimination_tree.cpp whatever the model says).
const ComputationModel ComputationModel::model_Hip_MI355X{
    // potrf: a + b n + c n^2 + d n^3
    {3.618102371671751e-08, 3.804901435864945e-07, 6.981640690980229e-11, 8.536479805489468e-16},
    // trsm: a + b n + c n^2 + (d + e n + f n^2) k
    {1.709462696726716e-07, 1.114928057617138e-07, 4.611165305099729e-10, 5.116797002889815e-10, 1.341879640359843e-11, 0.000000000000000e+00},
    // syge: a + b u + c v + k (d + e u + f v)
    {2.083655120359462e-07, 7.425127963497075e-10, 1.910711883161907e-12, 6.123296965670030e-08, 2.127243764724589e-12, 2.085578751310278e-15},
    // asmbl: a + b br + c bc + d br bc
    {1.819244623955457e-07, 0.000000000000000e+00, 1.854767815091107e-08, 1.095148499821501e-09}};

}  // namespace BaSpaCho
