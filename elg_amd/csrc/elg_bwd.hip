// Backward replay kernel (placeholder until the real kernel lands in this file).
#include "elg_rollout.h"
#include <string>
namespace elg { int fail(int code, const std::string& msg); }
extern "C" int elg_rollout_bwd(const elg_bwd_args* a, void* stream) {
    (void)a; (void)stream;
    return elg::fail(ELG_ENOTIMPL, "rollout_bwd not built yet");
}
