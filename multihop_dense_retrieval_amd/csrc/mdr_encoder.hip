// csrc/mdr_encoder.hip -- RoBERTa encoder forward (placeholder until the kernels land this round).
#include "mdr_common.h"

struct mdr_encoder { int unused; };

extern "C" {
int mdr_encoder_create(const mdr_encoder_config*, const mdr_tensor*, int, int, int, void*, mdr_encoder**) {
    return mdr::set_error(MDR_E_STATE, "encoder kernels not built yet");
}
int mdr_encoder_free(mdr_encoder*) { return MDR_OK; }
size_t mdr_encoder_workspace_bytes(const mdr_encoder*, int, int) { return 0; }
int mdr_encoder_forward(mdr_encoder*, const int64_t*, const int64_t*, int, int, float*, void*, size_t, void*) {
    return mdr::set_error(MDR_E_STATE, "encoder kernels not built yet");
}
}
