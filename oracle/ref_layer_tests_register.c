/* oracle/ref_layer_tests_register.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Glue for oracle/Makefile.layer_tests: the reference's own layer-validation programs
 * (the .cpp files of /root/reference/tests/validation_layer, compiled from where they lie with -DCSINN_API=14) are linked with the
 * genuine front-end (oracle/_ref/libshl_ref_x86.so) and with this repository's backend; somebody has to put the
 * backend into dispatch slot 14 before main() runs.  In tree that is one line of source/nn2/setup.c (INTEGRATION.md
 * option B); out of tree it is this constructor.  Registration only writes the two table slots
 * (source/nn2/setup.c:98-99,127-129), so it may run before the front-end's own lazy shl_init(). */
void shl_target_init_mi355x(void);

__attribute__((constructor)) static void register_mi355x_backend(void) { shl_target_init_mi355x(); }
