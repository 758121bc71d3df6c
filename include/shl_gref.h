/*
 * shl_gref.h -- the graph executor ("gref") surface a compute backend relies on.
 *
 * Restated from the reference's include/graph/shl_node.h:22-37 (struct shl_node, identical
 * field order: backends walk node lists built by either implementation), include/graph/
 * shl_gref.h (est callbacks) and source/graph_ref/setup.c:2031-2081 (runtime map).
 *
 * est callbacks have the operator's own signature and only RECORD the layer in the session's
 * graph; compute happens in csinn_session_run.
 */
#ifndef CSINN_MI355X_SHL_GREF_H_
#define CSINN_MI355X_SHL_GREF_H_

#include "shl_utils.h"

#ifdef __cplusplus
extern "C" {
#endif

/* node ids that are not operators (csinn_data_structure.h:333-336 of the reference) */
#define CSINN_TENSOR 195
#define CSINN_SUBGRAPH 196

struct shl_node {
    int type;              /* csinn_op_enum, CSINN_TENSOR or CSINN_SUBGRAPH */
    struct shl_node **in;
    struct shl_node **out;
    int subgraph_idx;
    int in_num;
    int out_num;
    char *name;
    void *data;            /* op node: params block; tensor node: struct csinn_tensor * */
    int ref_count;
    int ref_count_init;
    int visited;
    int *restricted_map;
    int restricted_map_num;
};

struct shl_node *shl_node_alloc(int node_type, char *name, int in_num, int out_num, void *data);
struct shl_node *shl_node_var_alloc(char *name, void *data);
struct shl_node *shl_node_const_var_alloc(char *name, void *data);
int shl_node_free(struct shl_node *node);
int shl_node_add_in(struct shl_node *node, struct shl_node *in, int index);
int shl_node_add_out(struct shl_node *node, struct shl_node *out, int index);

struct shl_ref_graph *shl_gref_get_graph(struct csinn_session *sess);
int shl_gref_graph_insert(struct shl_node *node, struct shl_ref_graph *graph);
/* calls fn with the operator's own argument list taken from the node's edges
 * (source/graph_ref/setup.c:75-268) */
int shl_gref_call_layer_func(void *fn, struct shl_node *node);
struct csinn_callback *shl_gref_best_callback(struct shl_node *node);

int shl_gref_conv2d(CSINN_CONV_ARGS);
int shl_gref_conv2d_relu(CSINN_CONV_ARGS);
int shl_gref_conv2d_relu6(CSINN_CONV_ARGS);
int shl_gref_depthwise_conv2d(CSINN_CONV_ARGS);
int shl_gref_depthwise_conv2d_relu(CSINN_CONV_ARGS);
int shl_gref_depthwise_conv2d_relu6(CSINN_CONV_ARGS);
int shl_gref_group_conv2d(CSINN_CONV_ARGS);
int shl_gref_group_conv2d_relu(CSINN_CONV_ARGS);
int shl_gref_group_conv2d_relu6(CSINN_CONV_ARGS);
int shl_gref_fullyconnected(struct csinn_tensor *input, struct csinn_tensor *output,
                            struct csinn_tensor *weights, struct csinn_tensor *bias,
                            struct csinn_fc_params *params);
int shl_gref_relu(struct csinn_tensor *input, struct csinn_tensor *output,
                  struct csinn_relu_params *params);
int shl_gref_relu6(struct csinn_tensor *input, struct csinn_tensor *output,
                   struct csinn_relu_params *params);
int shl_gref_add(struct csinn_tensor *input0, struct csinn_tensor *input1, struct csinn_tensor *output,
                 struct csinn_diso_params *params);
int shl_gref_global_avgpool2d(struct csinn_tensor *input, struct csinn_tensor *output,
                              struct csinn_pool_params *params);
int shl_gref_softmax(struct csinn_tensor *input, struct csinn_tensor *output,
                     struct csinn_softmax_params *params);

/* session-level handlers of the executor; a backend forwards the ones it does not override
 * (pattern: source/c920v2_opt/setup.c:355-389) */
void *shl_gref_runtime_callback(int runtime_op);
void shl_gref_session_init(struct csinn_session *sess);
void shl_gref_session_deinit(struct csinn_session *sess);
int shl_gref_session_setup(struct csinn_session *sess);
int shl_gref_session_run(struct csinn_session *sess);

#ifdef __cplusplus
}
#endif
#endif /* CSINN_MI355X_SHL_GREF_H_ */
