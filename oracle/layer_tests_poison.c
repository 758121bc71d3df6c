/* oracle/layer_tests_poison.c -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference's layer-mode test drivers hand the operator an output tensor that ALREADY holds the quantised expected
 * values (tests/validation_layer/testutil.h:870-879: qoutput = convert_f32_layer(output, ...) with output->data =
 * reference->data), so a backend that never wrote a byte would pass result_verify_f32.  oracle/Makefile.layer_tests
 * links the test programs with ld --wrap for the three operator entry points AS REFERENCED BY THE TEST'S OBJECT FILE:
 * the shim overwrites the output buffer with wildly alternating values (int8 +127 / -128,
 * binary16 1000 / 0: a CONSTANT fill still scores a cosine similarity of 0.99 against these all-positive expected
 * tensors) and then calls the real operator -- after which the reference's own verdict is a parity verdict.
 * (Seen with it: the reference's own x86 NCHW kernel fails its convolution test on this vector -- it computes image 0
 * of a batch only, SURVEY 0.5 -- and had been passing on the pre-filled expected values.)
 */
#include <string.h>

struct csinn_tensor;
int csinn_tensor_byte_size(struct csinn_tensor *tensor);

/* dtype: enum csinn_dtype_enum at offset 8 of struct csinn_tensor; CSINN_DTYPE_FLOAT16 = 8 */
static void poison(void *data, size_t bytes, int dtype)
{
    unsigned char *p = data;
    if (dtype == 8) {
        static const unsigned char pat[4] = {0xD0, 0x63, 0x00, 0x00}; /* 1000.0h, 0.0h (negative values turn the
                                                                         * reference's statistics into NaN, which it
                                                                         * does not count as a failure) */
        for (size_t i = 0; i < bytes; i++) p[i] = pat[i & 3];
    } else {
        for (size_t i = 0; i < bytes; i++) p[i] = (i & 1) ? 0x80 : 0x7F;
    }
}

#define WRAP5(name)                                                                                                  \
    int __real_##name(struct csinn_tensor *, struct csinn_tensor *, struct csinn_tensor *, struct csinn_tensor *, void *); \
    int __wrap_##name(struct csinn_tensor *in, struct csinn_tensor *out, struct csinn_tensor *k, struct csinn_tensor *b, \
                      void *params)                                                                                  \
    {                                                                                                                \
        void *data = *(void **)out; /* struct csinn_tensor: data at offset 0 (csinn_data_structure.h:505) */         \
        /* graph mode hands node pointers around in `data` while the graph is built: only poison real buffers, i.e.  \
         * layer mode, where the test allocated byte_size bytes with malloc (test_utils.c:647) */                    \
        extern int shl_layer_tests_graph_mode;                                                                       \
        if (data && !shl_layer_tests_graph_mode) poison(data, (size_t)csinn_tensor_byte_size(out), ((int *)out)[2]); \
        return __real_##name(in, out, k, b, params);                                                                 \
    }

int shl_layer_tests_graph_mode; /* set by -DLAYER_TESTS_GRAPH_MODE builds (fullyconnected.cpp runs a session) */
#ifdef LAYER_TESTS_GRAPH_MODE
__attribute__((constructor)) static void graph_mode(void) { shl_layer_tests_graph_mode = 1; }
#endif

WRAP5(csinn_conv2d)
WRAP5(csinn_depthwise_conv2d)
WRAP5(csinn_fullyconnected)
