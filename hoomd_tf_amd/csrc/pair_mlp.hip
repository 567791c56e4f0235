// placeholder until the MFMA pair-MLP evaluator lands (next commit)
#include "htf_common.h"
#include "htf_internal.h"
namespace htf {
struct MlpDevice { int unused; };
int mlp_create(const htf_potential_desc *, MlpDevice **) { set_error("pair-MLP evaluator is not built yet"); return HTF_ERR_INVALID; }
void mlp_destroy(MlpDevice *m) { delete m; }
int mlp_eval(const MlpDevice *, const void *, int, unsigned, unsigned, void *, int, hipStream_t) { set_error("pair-MLP evaluator is not built yet"); return HTF_ERR_INVALID; }
}
