// oracle/ref_wrap.cpp -- TEST INFRASTRUCTURE (never linked into the product library).
//
// extern "C" entry points, written for this repository, around the REAL reference (VTM 2.1) compiled
// from /root/reference by oracle/Makefile into oracle/_ref/libvtmref.so.  They let tests/, bench.py's
// cpu_baseline leg and the golden-vector generator call the reference's own kernels (scalar or the
// SIMD tables the reference itself would pick) through ctypes.  Every wrapper names the reference
// function it enters.  Nothing here re-implements reference arithmetic.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <iostream>

#include "CommonLib/CommonDef.h"
#include "CommonLib/Rom.h"
#include "Utilities/program_options_lite.h"
#include "EncApp.h"
#include "DecApp.h"

#include "ref_wrap_kernels.h"

extern "C" {

int vtmref_version(void) { return 21; }

// The reference's global tables (g_aucLog2, transform matrices, scan orders ...) are built by initROM()
// (CommonLib/Rom.cpp:206-460); EncLib::create / DecLib::create call it once (EncLib.cpp:86).  The kernel wrappers
// need the same state, so the library builds it when loaded.  initROM is idempotent-guarded here only.
static struct RomInit { RomInit() { initROM(); } } g_romInit;

// Runs the reference encoder application class exactly as App/EncoderApp/encmain.cpp:79-189 does
// (create -> parseCfg -> encode -> destroy).  argv[0] is ignored like a program name.
int vtmref_encode(int argc, char** argv)
{
  EncApp* app = new EncApp;
  app->create();
  try
  {
    if (!app->parseCfg(argc, argv)) { app->destroy(); return 1; }
  }
  catch (df::program_options_lite::ParseFailure& e)
  {
    std::cerr << "Error parsing option \"" << e.arg << "\" with argument \"" << e.val << "\"." << std::endl;
    return 1;
  }
  clock_t t0 = clock();
  try { app->encode(); }
  catch (Exception& e) { std::cerr << e.what() << std::endl; return 1; }
  catch (...) { std::cerr << "Unspecified error occurred" << std::endl; return 1; }
  clock_t t1 = clock();
  app->destroy();
  delete app;
  printf(" Total Time: %12.3f sec. [user]\n", (t1 - t0) * 1.0 / CLOCKS_PER_SEC);
  fflush(stdout);
  return 0;
}

// Runs the reference decoder application class as App/DecoderApp/decmain.cpp:53-131 does.
int vtmref_decode(int argc, char** argv)
{
  int rc = 0;
  DecApp* app = new DecApp;
  if (!app->parseCfg(argc, argv)) return 1;
  clock_t t0 = clock();
  try
  {
    if (0 != app->decode())
    {
      printf("\n\n***ERROR*** A decoding mismatch occured: signalled md5sum does not match\n");
      rc = 1;
    }
  }
  catch (Exception& e) { std::cerr << e.what() << std::endl; rc = 1; }
  catch (...) { std::cerr << "Unspecified error occurred" << std::endl; rc = 1; }
  printf("\n Total Time: %12.3f sec.\n", (double)(clock() - t0) / CLOCKS_PER_SEC);
  fflush(stdout);
  delete app;
  return rc;
}

}  // extern "C"
