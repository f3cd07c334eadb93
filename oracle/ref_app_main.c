/* oracle/ref_app_main.c -- TEST INFRASTRUCTURE.
 * Driver for the reference applications: `vtmref_app [--hip] enc <EncoderApp args>` / `vtmref_app [--hip] dec <DecoderApp args>`.
 * --hipsel loads libvtmref_hipsel.so: the reference built from a tree that carries integration/vtm-2.1-hip.patch (select with --SIMD=HIP).
 * --hip loads libvtmref_hip.so: the same reference objects with the in-loop filter entry points wrapped to the GPU shim.
 * The reference's own main() files need a cmake-generated header and are not built (oracle/Makefile);
 * the EncApp/DecApp classes are, inside libvtmref.so, and are entered through vtmref_encode/vtmref_decode
 * (oracle/ref_wrap.cpp), which follow App/EncoderApp/encmain.cpp:79-189 and App/DecoderApp/decmain.cpp. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <libgen.h>
#include <limits.h>
#include <stdlib.h>
typedef int (*appfn)(int, char**);
int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s [--hip] enc|dec <args...>\n", argv[0]); return 2; }
  char self[PATH_MAX]; if (!realpath(argv[0], self)) { perror("realpath"); return 2; }
  int hip = 0, sel = 0;
  if (strcmp(argv[1], "--hip") == 0) { hip = 1; argv++; argc--; if (argc < 2) return 2; }
  else if (strcmp(argv[1], "--hipsel") == 0) { sel = 1; argv++; argc--; if (argc < 2) return 2; }   /* the patched tree's library: pass --SIMD=HIP */
  const char* dir = dirname(self);     /* may modify `self`: called once */
  char lib[PATH_MAX]; snprintf(lib, sizeof lib, "%s/%s", dir, hip ? "libvtmref_hip.so" : sel ? "libvtmref_hipsel.so" : "libvtmref.so");
  void* hk = NULL;
  if (hip) {   /* interposes the two encoder-statistics entry points ld --wrap cannot reach (oracle/ref_hooks.cpp) */
    char hooks[PATH_MAX]; snprintf(hooks, sizeof hooks, "%s/libvtmhooks.so", dir);
    hk = dlopen(hooks, RTLD_NOW | RTLD_GLOBAL);
    if (!hk) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  }
  void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  if (hk) { void (*st)(void*) = (void (*)(void*))dlsym(hk, "vtmhooks_set_target"); if (!st) return 2; st(h); }
  appfn f = (appfn)dlsym(h, strcmp(argv[1], "enc") == 0 ? "vtmref_encode" : "vtmref_decode");
  if (!f) { fprintf(stderr, "dlsym: %s\n", dlerror()); return 2; }
  return f(argc - 1, argv + 1);
}
