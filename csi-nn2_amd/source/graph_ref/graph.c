/*
 * graph.c -- minimal sequential graph executor, a work-alike of the reference's "gref"
 * (source/graph_ref/) for the operators on the MI355X hot path.
 *
 * Model (source/graph_ref/utils.c:21-93, setup.c:617-797,1125-1154,1256-1450):
 *   - csinn_set_tensor_entry wraps a tensor in a CSINN_TENSOR node and parks the node in
 *     tensor->data until real data arrives through csinn_update_input;
 *   - every csinn_<op>() call in graph mode lands in an `est` callback below, which appends
 *     an operator node whose edges are tensor nodes;
 *   - session_setup resolves each layer's backend callbacks by params->api and runs `init`;
 *   - session_run walks the layers in insertion order: allocate outputs, `exec`, release
 *     inputs whose reference count dropped to zero.
 * Tensors handed to `exec` carry HOST pointers here, exactly as in the reference.  The
 * MI355X backend supplies its own device-resident SESSION_RUN on top of the same node list
 * (source/mi355x_opt/session.c).
 */
#include <string.h>

#include "shl_gref.h"

/* ------------------------------------------------------------------------ nodes
 * source/nn2/node.c:22-86 */
struct shl_node *shl_node_alloc(int node_type, char *name, int in_num, int out_num, void *data)
{
    struct shl_node *n = shl_mem_alloc(sizeof(struct shl_node));
    n->type = node_type;
    n->name = name;
    n->data = data;
    n->in_num = in_num;
    n->out_num = out_num;
    n->in = in_num ? shl_mem_alloc((int64_t)in_num * sizeof(struct shl_node *)) : NULL;
    n->out = out_num ? shl_mem_alloc((int64_t)out_num * sizeof(struct shl_node *)) : NULL;
    n->subgraph_idx = -1;
    return n;
}

struct shl_node *shl_node_var_alloc(char *name, void *data)
{
    return shl_node_alloc(CSINN_TENSOR, name, 1, 1, data);
}

struct shl_node *shl_node_const_var_alloc(char *name, void *data)
{
    return shl_node_alloc(CSINN_TENSOR, name, 0, 1, data);
}

int shl_node_free(struct shl_node *node)
{
    shl_mem_free(node->in);
    shl_mem_free(node->out);
    shl_mem_free(node);
    return CSINN_TRUE;
}

/* a tensor node records every consumer in its `out` list */
int shl_node_add_in(struct shl_node *node, struct shl_node *in, int index)
{
    node->in[index] = in;
    if (in->type != CSINN_TENSOR) return CSINN_TRUE;
    if (in->out_num == 1 && in->out[0] == NULL) {
        in->out[0] = node;
    } else {
        in->out = shl_mem_realloc(in->out, (size_t)(in->out_num + 1) * sizeof(struct shl_node *),
                                  (size_t)in->out_num * sizeof(struct shl_node *));
        in->out[in->out_num++] = node;
    }
    return CSINN_TRUE;
}

int shl_node_add_out(struct shl_node *node, struct shl_node *out, int index)
{
    node->out[index] = out;
    if (out->type == CSINN_TENSOR && out->in_num == 1) out->in[0] = node;
    return CSINN_TRUE;
}

/* ------------------------------------------------------------------------ graph */
struct shl_ref_graph *shl_gref_get_graph(struct csinn_session *sess)
{
    struct shl_gref_target_data *td = sess->td;
    return td ? td->graph : NULL;
}

int shl_gref_graph_insert(struct shl_node *node, struct shl_ref_graph *graph)
{
    if (graph->layer_index >= graph->layer_size) {
        int grown = graph->layer_size + 128;
        graph->layer = shl_mem_realloc(graph->layer, (size_t)grown * sizeof(struct shl_node *),
                                       (size_t)graph->layer_size * sizeof(struct shl_node *));
        graph->layer_size = grown;
    }
    graph->layer[graph->layer_index++] = node;
    return CSINN_TRUE;
}

/* single input, two constants, single output: conv / depthwise / fullyconnected */
static int record_sidcso(struct csinn_tensor *input, struct csinn_tensor *output,
                         struct csinn_tensor *const0, struct csinn_tensor *const1, int op,
                         void *params)
{
    struct csinn_params_base *base = params;
    struct shl_node *layer = shl_node_alloc(op, base->name, 3, 1, params);
    struct shl_node *produced = shl_node_var_alloc(output->name, output);
    shl_node_add_in(layer, (struct shl_node *)input->data, 0);
    shl_node_add_in(layer, shl_node_const_var_alloc(const0->name, const0), 1);
    shl_node_add_in(layer, shl_node_const_var_alloc(const1->name, const1), 2);
    shl_node_add_out(layer, produced, 0);
    output->data = produced;
    return shl_gref_graph_insert(layer, shl_gref_get_graph(input->sess));
}

static int record_siso(struct csinn_tensor *input, struct csinn_tensor *output, int op,
                       void *params)
{
    struct csinn_params_base *base = params;
    struct shl_node *layer = shl_node_alloc(op, base->name, 1, 1, params);
    struct shl_node *produced = shl_node_var_alloc(output->name, output);
    shl_node_add_in(layer, (struct shl_node *)input->data, 0);
    shl_node_add_out(layer, produced, 0);
    output->data = produced;
    return shl_gref_graph_insert(layer, shl_gref_get_graph(input->sess));
}

#define EST_CONV(fn, OP) \
    int fn(CSINN_CONV_ARGS) { return record_sidcso(input, output, kernel, bias, OP, params); }
EST_CONV(shl_gref_conv2d, CSINN_OP_CONV2D)
EST_CONV(shl_gref_conv2d_relu, CSINN_OP_CONV2D_RELU)
EST_CONV(shl_gref_conv2d_relu6, CSINN_OP_CONV2D_RELU6)
EST_CONV(shl_gref_depthwise_conv2d, CSINN_OP_DEPTHWISE_CONV2D)
EST_CONV(shl_gref_depthwise_conv2d_relu, CSINN_OP_DEPTHWISE_CONV2D_RELU)
EST_CONV(shl_gref_depthwise_conv2d_relu6, CSINN_OP_DEPTHWISE_CONV2D_RELU6)
EST_CONV(shl_gref_group_conv2d, CSINN_OP_GROUP_CONV2D)
EST_CONV(shl_gref_group_conv2d_relu, CSINN_OP_GROUP_CONV2D_RELU)
EST_CONV(shl_gref_group_conv2d_relu6, CSINN_OP_GROUP_CONV2D_RELU6)

int shl_gref_fullyconnected(struct csinn_tensor *input, struct csinn_tensor *output,
                            struct csinn_tensor *weights, struct csinn_tensor *bias,
                            struct csinn_fc_params *params)
{
    return record_sidcso(input, output, weights, bias, CSINN_OP_FULLYCONNECTED, params);
}

int shl_gref_relu(struct csinn_tensor *input, struct csinn_tensor *output,
                  struct csinn_relu_params *params)
{
    return record_siso(input, output, CSINN_OP_RELU, params);
}

int shl_gref_relu6(struct csinn_tensor *input, struct csinn_tensor *output,
                   struct csinn_relu_params *params)
{
    return record_siso(input, output, CSINN_OP_RELU6, params);
}

/* two activation inputs (either may also be a constant tensor), one output
 * (shl_gref_diso_op, source/graph_ref/utils.c of the reference) */
int shl_gref_add(struct csinn_tensor *input0, struct csinn_tensor *input1, struct csinn_tensor *output,
                 struct csinn_diso_params *params)
{
    struct shl_node *layer = shl_node_alloc(CSINN_OP_ADD, params->base.name, 2, 1, params);
    struct shl_node *produced = shl_node_var_alloc(output->name, output);
    struct csinn_tensor *ins[2] = {input0, input1};
    for (int i = 0; i < 2; i++) {
        if (ins[i]->is_const)
            shl_node_add_in(layer, shl_node_const_var_alloc(ins[i]->name, ins[i]), i);
        else
            shl_node_add_in(layer, (struct shl_node *)ins[i]->data, i);
    }
    shl_node_add_out(layer, produced, 0);
    output->data = produced;
    return shl_gref_graph_insert(layer, shl_gref_get_graph(output->sess ? output->sess : input0->sess));
}

int shl_gref_global_avgpool2d(struct csinn_tensor *input, struct csinn_tensor *output,
                              struct csinn_pool_params *params)
{
    return record_siso(input, output, CSINN_OP_GLOBAL_AVGPOOL2D, params);
}

int shl_gref_softmax(struct csinn_tensor *input, struct csinn_tensor *output,
                     struct csinn_softmax_params *params)
{
    return record_siso(input, output, CSINN_OP_SOFTMAX, params);
}

/* ------------------------------------------------------------------------ callbacks */
int shl_gref_call_layer_func(void *fn, struct shl_node *node)
{
    int (*f)() = fn;
    void *params = node->data;
    switch (node->type) {
        case CSINN_OP_RELU:
        case CSINN_OP_RELU6:
        case CSINN_OP_GLOBAL_AVGPOOL2D:
        case CSINN_OP_SOFTMAX:
            return f(node->in[0]->data, node->out[0]->data, params);
        case CSINN_OP_ADD:
            return f(node->in[0]->data, node->in[1]->data, node->out[0]->data, params);
        case CSINN_OP_CONV2D:
        case CSINN_OP_CONV2D_RELU:
        case CSINN_OP_CONV2D_RELU6:
        case CSINN_OP_DEPTHWISE_CONV2D:
        case CSINN_OP_DEPTHWISE_CONV2D_RELU:
        case CSINN_OP_DEPTHWISE_CONV2D_RELU6:
        case CSINN_OP_GROUP_CONV2D:
        case CSINN_OP_GROUP_CONV2D_RELU:
        case CSINN_OP_GROUP_CONV2D_RELU6:
        case CSINN_OP_FULLYCONNECTED:
            return f(node->in[0]->data, node->out[0]->data, node->in[1]->data, node->in[2]->data,
                     params);
        default:
            shl_debug_error("%s: op %d is outside this executor's scope\n", __func__, node->type);
            return CSINN_FALSE;
    }
}

/* pick the layer's callbacks from the backend named by params->api and the dtype of the
 * layer's first input (source/graph_ref/setup.c:617-654) */
struct csinn_callback *shl_gref_best_callback(struct shl_node *node)
{
    struct csinn_params_base *params = node->data;
    struct csinn_tensor *first = node->in[0]->data;
    shl_op_callback_map(params, node->type, first->dtype);
    return params->cb;
}

static struct csinn_callback g_est_only[24];

static struct csinn_callback *gref_cb_map(int op, int dtype)
{
    (void)dtype;
    static const struct { int op; int (*est)(); } table[] = {
        {CSINN_OP_CONV2D, shl_gref_conv2d},
        {CSINN_OP_CONV2D_RELU, shl_gref_conv2d_relu},
        {CSINN_OP_CONV2D_RELU6, shl_gref_conv2d_relu6},
        {CSINN_OP_DEPTHWISE_CONV2D, shl_gref_depthwise_conv2d},
        {CSINN_OP_DEPTHWISE_CONV2D_RELU, shl_gref_depthwise_conv2d_relu},
        {CSINN_OP_DEPTHWISE_CONV2D_RELU6, shl_gref_depthwise_conv2d_relu6},
        {CSINN_OP_GROUP_CONV2D, shl_gref_group_conv2d},
        {CSINN_OP_GROUP_CONV2D_RELU, shl_gref_group_conv2d_relu},
        {CSINN_OP_GROUP_CONV2D_RELU6, shl_gref_group_conv2d_relu6},
        {CSINN_OP_FULLYCONNECTED, shl_gref_fullyconnected},
        {CSINN_OP_RELU, shl_gref_relu},
        {CSINN_OP_RELU6, shl_gref_relu6},
        {CSINN_OP_GLOBAL_AVGPOOL2D, shl_gref_global_avgpool2d},
        {CSINN_OP_SOFTMAX, shl_gref_softmax},
        {CSINN_OP_ADD, shl_gref_add},
    };
    for (unsigned i = 0; i < sizeof(table) / sizeof(table[0]); i++) {
        if (table[i].op == op) {
            g_est_only[i].est = table[i].est;
            return &g_est_only[i];
        }
    }
    return NULL;
}

/* ------------------------------------------------------------------------ session handlers */
void shl_gref_session_init(struct csinn_session *sess)
{
    struct shl_gref_target_data *td = shl_mem_alloc(sizeof(struct shl_gref_target_data));
    td->graph = shl_mem_alloc(sizeof(struct shl_ref_graph));
    sess->td = td;
    sess->base_layout = CSINN_LAYOUT_NCHW; /* source/graph_ref/setup.c:66-73 */
}

void shl_gref_session_deinit(struct csinn_session *sess)
{
    struct shl_ref_graph *g = shl_gref_get_graph(sess);
    if (g) {
        shl_mem_free(g->input);
        shl_mem_free(g->output);
        shl_mem_free(g->layer);
        shl_mem_free(g);
    }
    shl_mem_free(sess->td);
    sess->td = NULL;
    shl_mem_free(sess->input);
    shl_mem_free(sess->output);
    sess->input = sess->output = NULL;
}

static void gref_set_input_number(int number, struct csinn_session *sess)
{
    struct shl_ref_graph *g = shl_gref_get_graph(sess);
    g->input_num = number;
    g->input = shl_mem_alloc((int64_t)number * sizeof(struct shl_node *));
}

static void gref_set_output_number(int number, struct csinn_session *sess)
{
    struct shl_ref_graph *g = shl_gref_get_graph(sess);
    g->output_num = number;
    g->output = shl_mem_alloc((int64_t)number * sizeof(struct shl_node *));
}

static int gref_set_tensor(struct csinn_tensor *t, struct csinn_session *sess)
{
    (void)sess;
    t->data = shl_node_var_alloc(t->name, t);
    return CSINN_TRUE;
}

static int gref_set_input(int index, struct csinn_tensor *t, struct csinn_session *sess)
{
    shl_gref_get_graph(sess)->input[index] = t->data;
    return CSINN_TRUE;
}

static int gref_set_output(int index, struct csinn_tensor *t, struct csinn_session *sess)
{
    struct shl_ref_graph *g = shl_gref_get_graph(sess);
    g->output[index] = t->is_const ? shl_node_const_var_alloc(t->name, t) : t->data;
    return CSINN_TRUE;
}

static int gref_get_input(int index, struct csinn_tensor *t, struct csinn_session *sess)
{
    csinn_tensor_copy(t, shl_gref_get_graph(sess)->input[index]->data);
    return CSINN_TRUE;
}

static int gref_get_output(int index, struct csinn_tensor *t, struct csinn_session *sess)
{
    csinn_tensor_copy(t, shl_gref_get_graph(sess)->output[index]->data);
    return CSINN_TRUE;
}

static int gref_update_input(int index, struct csinn_tensor *t, struct csinn_session *sess)
{
    struct csinn_tensor *slot = shl_gref_get_graph(sess)->input[index]->data;
    slot->data = t->data;
    return CSINN_TRUE;
}

static int gref_update_output(int index, struct csinn_tensor *t, struct csinn_session *sess)
{
    struct csinn_tensor *slot = shl_gref_get_graph(sess)->output[index]->data;
    slot->data = t->data;
    slot->mtype = CSINN_MEM_TYPE_CPU_ACC; /* caller-owned: never allocated/freed by run */
    return CSINN_TRUE;
}

int shl_gref_session_setup(struct csinn_session *sess)
{
    struct shl_ref_graph *g = shl_gref_get_graph(sess);
    int status = CSINN_TRUE;
    for (int i = 0; i < g->layer_index; i++) {
        struct shl_node *n = g->layer[i];
        struct csinn_params_base *params = n->data;
        n->subgraph_idx = i;
        /* callbacks are looked up as if in layer mode so that `init` is visible */
        int keep = sess->base_run_mode;
        sess->base_run_mode = CSINN_RM_LAYER;
        struct csinn_callback *cb = shl_gref_best_callback(n);
        sess->base_run_mode = keep;
        (void)params;
        if (cb->init != NULL && shl_gref_call_layer_func(cb->init, n) != CSINN_TRUE) {
            shl_debug_error("%s: init of layer %d (%s) failed\n", __func__, i,
                            n->name ? n->name : "?");
            status = CSINN_FALSE;
        }
    }
    /* reference counts: one per consumer, one for the producer, one more for graph outputs */
    for (int i = 0; i < g->layer_index; i++) {
        struct shl_node *n = g->layer[i];
        for (int j = 0; j < n->in_num; j++)
            if (n->in[j]->ref_count_init > 0) n->in[j]->ref_count_init++;
        for (int k = 0; k < n->out_num; k++) n->out[k]->ref_count_init++;
    }
    for (int i = 0; i < g->output_num; i++) g->output[i]->ref_count_init++;
    return status;
}

int shl_gref_session_run(struct csinn_session *sess)
{
    struct shl_ref_graph *g = shl_gref_get_graph(sess);
    int status = CSINN_TRUE;
    for (int i = 0; i < g->layer_index; i++)
        for (int k = 0; k < g->layer[i]->out_num; k++)
            g->layer[i]->out[k]->ref_count = g->layer[i]->out[k]->ref_count_init;

    for (int i = 0; i < g->layer_index; i++) {
        struct shl_node *n = g->layer[i];
        struct csinn_params_base *params = n->data;
        for (int k = 0; k < n->out_num; k++) {
            struct csinn_tensor *t = n->out[k]->data;
            if (t->mtype != CSINN_MEM_TYPE_CPU_ACC) t->data = shl_mem_alloc(csinn_tensor_byte_size(t));
        }
        if (params->cb->exec == NULL ||
            shl_gref_call_layer_func(params->cb->exec, n) != CSINN_TRUE) {
            shl_debug_error("%s: layer %d (%s) failed\n", __func__, i, n->name ? n->name : "?");
            status = CSINN_FALSE;
        }
        for (int j = 0; j < n->in_num; j++) {
            struct shl_node *in = n->in[j];
            if (in->ref_count > 0 && --in->ref_count == 0) {
                struct csinn_tensor *t = in->data;
                if (t->mtype != CSINN_MEM_TYPE_CPU_ACC && csinn_tensor_size(t) != 0)
                    shl_mem_free(t->data);
            }
        }
        for (int k = 0; k < n->out_num; k++) n->out[k]->ref_count--;
    }
    return status;
}

void *shl_gref_runtime_callback(int op)
{
    switch (op) {
        case CSINN_SESSION_INIT: return shl_gref_session_init;
        case CSINN_SESSION_DEINIT: return shl_gref_session_deinit;
        case CSINN_SESSION_SETUP: return shl_gref_session_setup;
        case CSINN_SESSION_RUN: return shl_gref_session_run;
        case CSINN_UPDATE_INPUT: return gref_update_input;
        case CSINN_UPDATE_OUTPUT: return gref_update_output;
        case CSINN_SET_INPUT_NUMBER: return gref_set_input_number;
        case CSINN_SET_OUTPUT_NUMBER: return gref_set_output_number;
        case CSINN_SET_INPUT: return gref_set_input;
        case CSINN_SET_OUTPUT: return gref_set_output;
        case CSINN_GET_INPUT: return gref_get_input;
        case CSINN_GET_OUTPUT: return gref_get_output;
        case CSINN_TENSOR_ENTRY: return gref_set_tensor;
        default: return NULL;
    }
}

void shl_target_init_gref(void)
{
    shl_register_runtime_callback(CSINN_GREF, shl_gref_runtime_callback);
    shl_register_op_callback(CSINN_GREF, gref_cb_map);
}
