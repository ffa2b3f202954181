// oracle/ref_bind.cpp -- TEST INFRASTRUCTURE.  A binding translation unit of OURS that #includes the
// reference's header-only penalty functions (penalty_functions/*.h, found through -I$(REF) at build
// time, never copied) and exposes their scalar and v4sf overloads with C linkage so tests can pin
// oracle/slowflow_oracle.c's orc_psi_deriv_* against the real thing.  It defines no stand-in for
// anything the reference needs: those headers only depend on <math.h> and <xmmintrin.h>.
#include "penalty_functions/penalty_function.h"
#include "penalty_functions/quadratic_function.h"
#include "penalty_functions/modified_l1_norm.h"
#include "penalty_functions/lorentzian.h"
#include "penalty_functions/trunc_modified_l1_norm.h"
#include "penalty_functions/geman_mcclure.h"

// same id -> class map as Variational_AUX_MT::select_robust_function (variational_aux_mt.cpp:909-925)
static PenaltyFunction *make(int id, float eps, float trunc) {
    switch (id) {
    case 0: return new QuadraticFunction();
    case 2: return new Lorentzian(eps);
    case 3: return new TruncModifiedL1Norm(eps, trunc);
    case 4: return new GemanMcClure(eps);
    default: return new ModifiedL1Norm(eps);
    }
}

extern "C" void ref_penalty_derivative(int id, float eps, float trunc, const float *xsq, int n,
                                       float *out_scalar, float *out_vec) {
    PenaltyFunction *f = make(id, eps, trunc);
    for (int i = 0; i < n; i++) out_scalar[i] = f->derivative(xsq[i]);
    for (int i = 0; i + 4 <= n; i += 4) {
        v4sf x = {xsq[i], xsq[i + 1], xsq[i + 2], xsq[i + 3]};
        v4sf y = f->derivative(x);
        out_vec[i] = y[0]; out_vec[i + 1] = y[1]; out_vec[i + 2] = y[2]; out_vec[i + 3] = y[3];
    }
    delete f;
}

// psi itself (apply), the v4sf overload optimizeOcc evaluates (variational_aux_mt.cpp:819-830)
extern "C" void ref_penalty_apply(int id, float eps, float trunc, const float *xsq, int n, float *out_vec) {
    PenaltyFunction *f = make(id, eps, trunc);
    for (int i = 0; i + 4 <= n; i += 4) {
        v4sf x = {xsq[i], xsq[i + 1], xsq[i + 2], xsq[i + 3]};
        v4sf y = f->apply(x);
        out_vec[i] = y[0]; out_vec[i + 1] = y[1]; out_vec[i + 2] = y[2]; out_vec[i + 3] = y[3];
    }
    delete f;
}
