/* oracle/example_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Glue for oracle/Makefile.example: the reference's model example example/c906_mobilenetv1_f16.c is compiled from
 * where it lies under /root/reference, UNCHANGED, and linked with the genuine front-end + graph executor
 * (oracle/_ref/libshl_ref_x86.so) and with one of two occupants of the dispatch slot the example hard-codes
 * (`sess->base_api = CSINN_C906`, c906_mobilenetv1_f16.c:24 -- a slot the x86 build of libshl leaves empty):
 *
 *   -DHARNESS_MI355X   this repository's backend (shl_target_init_mi355x_slot(CSINN_C906))
 *   (default)          the reference's own C kernels: shl_cb_map_ref + shl_gref_runtime_callback in that slot
 *
 * The example runs on whatever malloc() hands it ("alloc random params", :1958-1964): quantisation records, weights
 * and the image are uninitialised memory.  To make the two runs comparable the link wraps three symbols AS REFERENCED
 * BY THE EXAMPLE'S OBJECT FILE ONLY (ld --wrap; the libraries' own malloc calls are untouched):
 *
 *   malloc(8453888)            <- the bytes of $SHL_EXAMPLE_PARAMS   (tests/test_ref_example.py writes them)
 *   malloc(224*224*3*2)        <- the bytes of $SHL_EXAMPLE_INPUT
 *   csinn_session_deinit(sess) -> first dumps graph output 0 (1000 binary16 values) to $SHL_EXAMPLE_OUTPUT and says
 *                                 how the session executed, then calls the real one
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct csinn_session;
struct csinn_tensor;
struct csinn_callback;
void shl_register_op_callback(int api, void *cb);
void shl_register_runtime_callback(int api, void *cb);
struct csinn_tensor *csinn_alloc_tensor(struct csinn_session *sess);
int csinn_get_output(int index, struct csinn_tensor *output, struct csinn_session *sess);
int csinn_tensor_byte_size(struct csinn_tensor *tensor);

#define SLOT_C906 3 /* enum csinn_api_enum, include/csinn/csinn_data_structure.h:98 */

#ifdef HARNESS_MI355X
int shl_target_init_mi355x_slot(int api);
int shl_mi355x_session_is_device_resident(struct csinn_session *sess);
int shl_mi355x_session_fused_pairs(struct csinn_session *sess);
int shl_mi355x_session_folded_activations(struct csinn_session *sess);
__attribute__((constructor)) static void occupy_slot(void) { shl_target_init_mi355x_slot(SLOT_C906); }
#else
struct csinn_callback *shl_cb_map_ref(int op, int dtype);
void *shl_gref_runtime_callback(int op);
__attribute__((constructor)) static void occupy_slot(void)
{
    shl_register_op_callback(SLOT_C906, shl_cb_map_ref);
    shl_register_runtime_callback(SLOT_C906, shl_gref_runtime_callback);
}
#endif

void *__real_malloc(size_t n);
void __real_csinn_session_deinit(struct csinn_session *sess);

static void fill_from(const char *env, void *p, size_t n)
{
    const char *path = getenv(env);
    if (path == NULL || *path == 0) return;
    FILE *f = fopen(path, "rb");
    if (f == NULL || fread(p, 1, n, f) != n) {
        fprintf(stderr, "example_harness: cannot read %zu bytes from %s=%s\n", n, env, path);
        exit(3);
    }
    fclose(f);
}

void *__wrap_malloc(size_t n)
{
    void *p = __real_malloc(n);
    if (p == NULL) return p;
    if (n == 8453888) fill_from("SHL_EXAMPLE_PARAMS", p, n);
    else if (n == 224 * 224 * 3 * 2) fill_from("SHL_EXAMPLE_INPUT", p, n);
    return p;
}

void __wrap_csinn_session_deinit(struct csinn_session *sess)
{
    /* struct csinn_tensor: data at offset 0 (csinn_data_structure.h:505-520) */
    struct csinn_tensor *out = csinn_alloc_tensor(NULL);
    csinn_get_output(0, out, sess);
    void *data = *(void **)out;
    const int bytes = csinn_tensor_byte_size(out);
    const char *path = getenv("SHL_EXAMPLE_OUTPUT");
    if (path && *path && data) {
        FILE *f = fopen(path, "wb");
        if (f == NULL || fwrite(data, 1, (size_t)bytes, f) != (size_t)bytes) {
            fprintf(stderr, "example_harness: cannot write %s\n", path);
            exit(3);
        }
        fclose(f);
    }
#ifdef HARNESS_MI355X
    printf("example_harness: output %d bytes, device_resident=%d fused_pairs=%d folded_activations=%d\n", bytes,
           shl_mi355x_session_is_device_resident(sess), shl_mi355x_session_fused_pairs(sess),
           shl_mi355x_session_folded_activations(sess));
#else
    printf("example_harness: output %d bytes, reference kernels\n", bytes);
#endif
    fflush(stdout);
    __real_csinn_session_deinit(sess);
}
