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
#include "EncoderLib/EncSlice.h"
#include "EncoderLib/EncCu.h"

// ---- chunk hand-over side record: the encoder state OUTSIDE the decoded picture buffer that a worker starting at an
// intra-period boundary needs to reproduce the sequential encoder -- the per-temporal-layer ATMVP statistics of EncCu (EncCu.h:119-122, filled by
// CABACWriter.cpp:543-552, consumed by EncSlice.cpp:1250-1306).  EncGOP calls EncSlice::compressSlice from another translation unit, so GNU ld
// --wrap reaches it (oracle/Makefile).  Environment (tools/chunk_exactness.py, tests/test_chunk_stitch.py):
//   VVCGPU_ATMVP_POC=P     the boundary: the record belongs to the slice with POC == P (the first picture the re-entered run codes itself;
//                          pictures before it IN CODING ORDER, e.g. POC 48/40/36/34 for P = 33, are decoded from the sequential stream)
//   VVCGPU_ATMVP_DUMP=file the sequential run writes the record (22 uint32: size[10], num[10], prevPOC, clear flag) when it reaches that slice
//   VVCGPU_ATMVP_LOAD=file the re-entered run installs the record before it codes that slice
extern "C" void __real__ZN8EncSlice13compressSliceEP7Picturebb(EncSlice* self, Picture* pic, bool entire, bool fastDqp);
extern "C" void __wrap__ZN8EncSlice13compressSliceEP7Picturebb(EncSlice* self, Picture* pic, bool entire, bool fastDqp)
{
  static int done = 0;
  const char* pocS = getenv("VVCGPU_ATMVP_POC");
  if (!done && pocS && pic->getPOC() == atoi(pocS))
  {
    done = 1;
    EncCu* cu = self->m_pcCuEncoder;
    unsigned rec[22];
    if (const char* f = getenv("VVCGPU_ATMVP_DUMP"))
    {
      for (int i = 0; i < 10; i++) { rec[i] = cu->m_subMergeBlkSize[i]; rec[10 + i] = cu->m_subMergeBlkNum[i]; }
      rec[20] = cu->m_prevPOC; rec[21] = cu->m_clearSubMergeStatic ? 1u : 0u;
      if (FILE* fp = fopen(f, "wb")) { fwrite(rec, sizeof rec, 1, fp); fclose(fp); }
    }
    if (const char* f = getenv("VVCGPU_ATMVP_LOAD"))
    {
      FILE* fp = fopen(f, "rb");
      if (fp && fread(rec, sizeof rec, 1, fp) == 1)
      {
        for (int i = 0; i < 10; i++) { cu->m_subMergeBlkSize[i] = rec[i]; cu->m_subMergeBlkNum[i] = rec[10 + i]; }
        cu->m_prevPOC = rec[20]; cu->m_clearSubMergeStatic = rec[21] != 0;
      }
      if (fp) fclose(fp);
    }
  }
  if (getenv("VVCGPU_ATMVP_TRACE"))
  {
    EncCu* cu = self->m_pcCuEncoder;
    fprintf(stderr, "[atmvp] poc %d prev %d clear %d num", pic->getPOC(), cu->m_prevPOC, (int)cu->m_clearSubMergeStatic);
    for (int i = 0; i < 6; i++) fprintf(stderr, " %u/%u", cu->m_subMergeBlkSize[i], cu->m_subMergeBlkNum[i]);
    fprintf(stderr, "\n");
  }
  __real__ZN8EncSlice13compressSliceEP7Picturebb(self, pic, entire, fastDqp);
}

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
  // the SIMD selector is read before anything else, as App/EncoderApp/encmain.cpp:97-105 does (the tables are filled by constructors); the library
  // built from a tree that carries integration/vtm-2.1-hip.patch accepts --SIMD=HIP here
  for (int i = 1; i < argc; i++)
    if (!strncmp(argv[i], "--SIMD=", 7)) read_x86_extension_flags(std::string(argv[i] + 7));
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
  for (int i = 1; i < argc; i++)                            // the decoder application has no SIMD option of its own: taken out of the list here
    if (!strncmp(argv[i], "--SIMD=", 7))
    {
      read_x86_extension_flags(std::string(argv[i] + 7));
      for (int k = i; k + 1 < argc; k++) argv[k] = argv[k + 1];
      argc--; i--;
    }
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
