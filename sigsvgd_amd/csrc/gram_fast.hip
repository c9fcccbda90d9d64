#include "sig_common.h"
namespace sigsvgd {
bool fast_supported(int, int, int, int, int, int, unsigned) { return false; }
int fast_workspace_bytes(int, int, int, int, int, unsigned, size_t *bytes) { *bytes = 0; return SIGSVGD_OK; }
int fast_launch(const GramProblem &) { set_error("fast path not built"); return SIGSVGD_E_UNSUPPORTED; }
}
