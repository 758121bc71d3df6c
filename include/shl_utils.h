/*
 * shl_utils.h -- the plug-in surface of the front-end: backend registration,
 * callback lookup, memory and logging helpers.
 *
 * Restated from the reference's include/shl_utils.h:39-86, shl_memory.h and
 * shl_debug.h; behaviour follows source/nn2/setup.c:98-147 (dispatch tables),
 * source/nn2/utils.c:2316-2352 (callback pickers), source/utils/memory.c:62-178
 * (zeroing allocator, weak symbols) and source/utils/debug.c:31-103.
 */
#ifndef CSINN_MI355X_SHL_UTILS_H_
#define CSINN_MI355X_SHL_UTILS_H_

#include <stdint.h>
#include <stdlib.h>

#include "csinn/csi_nn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- backend registration (source/nn2/setup.c:98-99,127-129) ---------------
 * op map     : struct csinn_callback *(*)(int op, int dtype)
 * runtime map: void *(*)(int runtime_op)   (enum csinn_runtime_enum -> handler) */
void shl_register_op_callback(int api, void *cb);
void shl_register_runtime_callback(int api, void *cb);
/* copies the backend's callback record for (base->api, op, dtype) into base->cb */
int shl_op_callback_map(struct csinn_params_base *base, int op, int dtype);
void *shl_get_runtime_callback(struct csinn_session *sess, int op);

/* init callback to run now (NULL in graph modes) / highest-priority compute cb */
void *shl_get_init_cb(struct csinn_params_base *base);
void *shl_get_p0_cb(struct csinn_params_base *base);
enum csinn_rmode_enum shl_get_run_mode(struct csinn_params_base *base);

/* ---- zero-filling allocator -------------------------------------------------- */
void *shl_mem_alloc(int64_t size);
void *shl_mem_calloc(size_t nmemb, size_t size);
void *shl_mem_realloc(void *ptr, size_t size, size_t orig_size);
void shl_mem_free(void *ptr);

/* ---- logging, gated by shl_debug_set_level() ---------------------------------- */
void shl_debug_debug(const char *format, ...);
void shl_debug_info(const char *format, ...);
void shl_debug_warning(const char *format, ...);
void shl_debug_error(const char *format, ...);
void shl_debug_fatal(const char *format, ...);
int shl_debug_get_level();
void shl_debug_set_level(int level);

/* monotonic nanoseconds (source/nn2/utils.c:2360-2365) */
uint64_t shl_get_timespec();

/* ---- graph-executor state reachable from a session (include/shl_utils.h:43-57) */
struct shl_node;
struct shl_ref_graph {
    struct shl_node **input;
    struct shl_node **output;
    int input_num;
    int output_num;
    struct shl_node **layer;
    int layer_size;
    int layer_index;
};

struct shl_gref_target_data {
    struct shl_ref_graph *graph;
    int is_hybrid_quantization_type;
    void *cpu_option;
};

#ifdef __cplusplus
}
#endif
#endif /* CSINN_MI355X_SHL_UTILS_H_ */
