/*
 * csinn_runtime.h -- session / tensor / params life-cycle of the CSI-NN2 API.
 *
 * Restated prototypes (own wording) of the subset of the reference's
 * include/csinn/csinn_runtime.h:79-340 that a user of conv2d /
 * depthwise_conv2d / fullyconnected touches.  Semantics follow
 * source/nn2/utils.c:308-484 and source/nn2/setup.c:77-514 of the reference.
 */
#ifndef CSINN_MI355X_RUNTIME_H_
#define CSINN_MI355X_RUNTIME_H_

#include "csinn_data_structure.h"

#ifdef __cplusplus
extern "C" {
#endif

/* element count (0 when dim_count == 0) and byte size of a tensor */
int csinn_tensor_size(struct csinn_tensor *tensor);
int csinn_tensor_byte_size(struct csinn_tensor *tensor);

/* zero-initialised tensor with one quant record {zp 0, scale 1}; inherits
 * dtype/layout/sess from `session` when it is not NULL */
struct csinn_tensor *csinn_alloc_tensor(struct csinn_session *session);
void csinn_free_tensor(struct csinn_tensor *tensor);
void csinn_realloc_quant_info(struct csinn_tensor *tensor, int quant_info_num);
/* shallow copy: shares `data`, duplicates the quant records */
void csinn_tensor_copy(struct csinn_tensor *dest, struct csinn_tensor *src);
/* dtype conversion (same layout) through the tensors' quant records */
int csinn_tensor_data_convert(struct csinn_tensor *dest, struct csinn_tensor *src);

/* zeroed params block of `params_size` bytes whose base.cb points at a fresh
 * struct csinn_callback; api/layout/quant_type/sess come from the session */
void *csinn_alloc_params(int params_size, struct csinn_session *session);
void csinn_free_params(void *params);

struct csinn_session *csinn_alloc_session();
void csinn_free_session(struct csinn_session *session);
void csinn_session_init(struct csinn_session *session);
void csinn_session_deinit(struct csinn_session *session);
int csinn_session_setup(struct csinn_session *session);
int csinn_session_run(struct csinn_session *session);

void csinn_set_input_number(int number, struct csinn_session *sess);
void csinn_set_output_number(int number, struct csinn_session *sess);
int csinn_get_input_number(struct csinn_session *sess);
int csinn_get_output_number(struct csinn_session *sess);
int csinn_set_input(int index, struct csinn_tensor *input, struct csinn_session *sess);
int csinn_set_output(int index, struct csinn_tensor *output, struct csinn_session *sess);
int csinn_get_input(int index, struct csinn_tensor *input, struct csinn_session *sess);
int csinn_get_output(int index, struct csinn_tensor *output, struct csinn_session *sess);
int csinn_update_input(int index, struct csinn_tensor *input, struct csinn_session *sess);
int csinn_update_output(int index, struct csinn_tensor *output, struct csinn_session *sess);
int csinn_set_tensor_entry(struct csinn_tensor *tensor, struct csinn_session *sess);

#ifdef __cplusplus
}
#endif
#endif /* CSINN_MI355X_RUNTIME_H_ */
