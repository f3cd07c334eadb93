// oracle/restate/intra.cpp -- TEST INFRASTRUCTURE: scalar restatement of the intra sample predictors (next row N4).
//   IntraPrediction::predIntraAng (mode switch + simplified PDPC, JVET_K0063)      CommonLib/IntraPrediction.cpp:251-347
//   IntraPrediction::xGetPredValDc (JVET_K0122) / xPredIntraDc                      :173-211, :482-493
//   IntraPrediction::getWideAngle / setReferenceArrayLengths (JVET_K0500)           :213-249
//   IntraPrediction::xPredIntraPlanar                                               :424-477
//   IntraPrediction::xPredIntraAng (HM_4TAPIF_AS_IN_JEM: linear filter iff deltaFract != 0; angular PDPC)   :540-773
//   IntraPrediction::xFilterReferenceSamples (regular [1 2 1] filter; strong smoothing is HEVC_TOOLS only)  :1006-1105
// Pinned against the compiled reference's own IntraPrediction::predIntraAng / xFilterReferenceSamples by tests/golden/intra.npz.
//
// Reference samples are handed over packed: refs[0] = top-left, refs[1 .. topLen] = row above (left to right),
// refs[topLen + 1 .. topLen + leftLen] = column to the left (top to bottom); the reference keeps the same values in a 2-D
// buffer (row 0 = top, column 0 = left, :262-263).
#include "orc_common.h"

namespace {

const int kAng[27]    = { 0, 1, 2, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 26, 29, 32, 35, 39, 45, 49, 54, 60, 68, 79, 93, 114 };
const int kInvAng[27] = { 0, 8192, 4096, 2731, 1638, 1170, 910, 745, 630, 546, 482, 431, 390, 356, 315, 282, 256, 234, 210, 182, 167, 152, 137, 120, 104, 88, 72 };
enum { PLANAR = 0, DC = 1, HOR = 18, DIA = 34, VER = 50, VDIA = 66 };

int ilog2(int v) { int l = 0; while ((1 << (l + 1)) <= v) l++; return l; }
Pel clipPel(int v, int lo, int hi) { return (Pel)(v < lo ? lo : (v > hi ? hi : v)); }

}  // namespace

ORC_API void orc_intra_ref_lengths(int w, int h, int* topLen, int* leftLen)       // :233-249
{
  const int ratio = std::min(2, std::abs(ilog2(w) - ilog2(h)));
  *leftLen = h << 1; *topLen = w << 1;
  if (w > h) *leftLen += (w >> ratio) - h + ((w + 31) >> 5);
  else if (h > w) *topLen += (h >> ratio) - w + ((h + 31) >> 5);
}

ORC_API void orc_intra_filter_refs(const Pel* in, Pel* out, int w, int h)         // :1071-1104, packed layout
{
  int T, L; orc_intra_ref_lengths(w, h, &T, &L);
  auto left = [&](const Pel* r, int i) { return i == 0 ? r[0] : r[T + i]; };      // left(0) = top-left
  out[T + L] = in[T + L];                                                         // bottom left, not filtered
  for (int i = L - 1; i >= 1; i--) out[T + i] = (Pel)((left(in, i + 1) + 2 * left(in, i) + left(in, i - 1) + 2) >> 2);
  out[0] = (Pel)((left(in, 1) + 2 * in[0] + in[1] + 2) >> 2);
  for (int i = 1; i < T; i++) out[i] = (Pel)((in[i + 1] + 2 * in[i] + in[i - 1] + 2) >> 2);
  out[T] = in[T];                                                                 // top right, not filtered
}

ORC_API int orc_intra_pred(const Pel* refs, Pel* dst, int dstStride, int w, int h, int dirMode, int clpMin, int clpMax)
{
  int T, L; orc_intra_ref_lengths(w, h, &T, &L);
  const Pel* top = refs;                                        // top[0] = top-left, top[1 + x]
  auto left = [&](int i) -> int { return i == 0 ? refs[0] : refs[T + i]; };
  const int log2W = ilog2(w), log2H = ilog2(h);
  const int scale = (log2W - 2 + log2H - 2 + 2) >> 2;

  if (dirMode == PLANAR)                                         // :424-477
  {
    const int bottomLeft = left(h + 1), topRight = top[w + 1];
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
      {
        const int horPred = (left(y + 1) << log2W) + (x + 1) * (topRight - left(y + 1));
        const int vertPred = (top[x + 1] << log2H) + (y + 1) * (bottomLeft - top[x + 1]);
        dst[y * dstStride + x] = (Pel)(((horPred << log2H) + (vertPred << log2W) + w * h) >> (1 + log2W + log2H));
      }
  }
  else if (dirMode == DC)                                        // :173-211
  {
    const int denom = (w == h) ? (w << 1) : std::max(w, h);
    int sum = 0;
    if (w >= h) for (int i = 0; i < w; i++) sum += top[1 + i];
    if (w <= h) for (int i = 0; i < h; i++) sum += left(1 + i);
    const Pel dc = (Pel)((sum + (denom >> 1)) >> ilog2(denom));
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) dst[y * dstStride + x] = dc;
  }
  else
  {
    // :545-773
    int predMode = dirMode;                                      // getWideAngle :213-231
    {
      const int modeShift = (std::min(2, std::abs(log2W - log2H)) << 2) + 2;
      if (w > h && predMode < 2 + modeShift) predMode += VDIA - 1;
      else if (h > w && predMode > VDIA - modeShift) predMode -= VDIA - 1;
    }
    const bool isVer = predMode >= DIA;
    const int angMode = isVer ? predMode - VER : -(predMode - HOR);
    const int absAngMode = std::abs(angMode);
    const int invAngle = kInvAng[absAngMode], angle = (angMode < 0 ? -1 : 1) * kAng[absAngMode];
    // main / side reference as functions of a signed index; index 0 = top-left
    const int W = isVer ? w : h, H = isVer ? h : w;              // block in the orientation of the main reference
    auto mainRef = [&](int i) -> int { return isVer ? top[i] : left(i); };
    auto sideRef = [&](int i) -> int { return isVer ? left(i) : top[i]; };
    auto refMain = [&](int k) -> int                             // :588-609: negative indices are projected from the side reference
    {
      if (k >= 0 || angle >= 0) return mainRef(k);
      return sideRef((128 + (-k) * invAngle) >> 8);
    };
    const int sideLen = isVer ? L : T;
    for (int y = 0; y < H; y++)
    {
      const int deltaPos = (y + 1) * angle, deltaInt = deltaPos >> 5, deltaFract = deltaPos & 31;
      for (int x = 0; x < W; x++)
      {
        int v;
        if (angle == 0) v = refMain(x + 1);
        else if (deltaFract) v = ((32 - deltaFract) * refMain(x + deltaInt + 1) + deltaFract * refMain(x + deltaInt + 2) + 16) >> 5;
        else v = refMain(x + deltaInt + 1);
        v = (Pel)v;
        if (angle != 0)                                          // angular PDPC :690-744 (inside the non-zero-angle branch)
        {
          if (predMode == 2 || predMode == VDIA)
          {
            const int wT = 16 >> std::min(31, (y << 1) >> scale), wL = 16 >> std::min(31, (x << 1) >> scale);
            if (wT + wL != 0)
            {
              const int c = x + y + 1;
              const int l = wL != 0 ? sideRef(c + 1) : 0, t = wT != 0 ? mainRef(c + 1) : 0;
              v = clipPel((wL * l + wT * t + (64 - wL - wT) * v + 32) >> 6, clpMin, clpMax);
            }
          }
          else if ((predMode >= VDIA - 8 && predMode != VDIA) || (predMode != 2 && predMode <= 2 + 8))
          {
            const int deltaPos0 = (2 + (x + 1) * invAngle) >> 2, deltaFrac0 = deltaPos0 & 63, deltaInt0 = deltaPos0 >> 6;
            const int deltay = y + deltaInt0 + 1;
            const int wL = 32 >> std::min(31, (x << 1) >> scale);
            if (deltay <= sideLen - 1 && wL != 0)
            {
              const int l = ((64 - deltaFrac0) * sideRef(deltay) + deltaFrac0 * sideRef(deltay + 1) + 32) >> 6;
              v = clipPel((wL * (Pel)l + (64 - wL) * v + 32) >> 6, clpMin, clpMax);
            }
          }
        }
        if (isVer) dst[y * dstStride + x] = (Pel)v; else dst[x * dstStride + y] = (Pel)v;
      }
    }
  }

  // simplified PDPC for planar / DC / horizontal / vertical :290-347
  if (dirMode == PLANAR || dirMode == DC || dirMode == HOR || dirMode == VER)
  {
    const int topLeft = top[0];
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
      {
        const int wT = 32 >> std::min(31, (y << 1) >> scale), wL = 32 >> std::min(31, (x << 1) >> scale);
        const int l = left(y + 1), t = top[x + 1], p = dst[y * dstStride + x];
        int v;
        if (dirMode == PLANAR) v = (wL * l + wT * t + (64 - wL - wT) * p + 32) >> 6;
        else if (dirMode == DC) { const int wTL = (wL >> 4) + (wT >> 4); v = (wL * l + wT * t - wTL * topLeft + (64 - wL - wT + wTL) * p + 32) >> 6; }
        else if (dirMode == HOR) v = (wT * t - wT * topLeft + 64 * p + 32) >> 6;
        else v = (wL * l - wL * topLeft + 64 * p + 32) >> 6;
        dst[y * dstStride + x] = clipPel(v, clpMin, clpMax);
      }
  }
  return 0;
}

// ---- reference sample gathering: IntraPrediction::xFillReferenceSamples :807-1004 ----------------------------------------------
// flags[0 .. totalUnits): the reference's neighborFlags, chain order bottom-left ... left ... top-left ... above ... above-right
// (left units are uh samples high, the top-left and above units uw samples wide).  rec points at the block's top-left sample in
// the reconstruction.  Output packed like the predictors' input (refs[0] top-left, then T above, then L left).
// The reference pads through a line buffer: an unavailable unit repeats the last sample of the unit before it (after that one
// was padded itself), and a leading run of unavailable units repeats the first sample of the first available unit.  In closed
// form: every sample of an unavailable unit equals one fixed sample of the nearest available unit.
ORC_API int orc_intra_fill_refs(const Pel* rec, int recStride, const uint8_t* flags, Pel* refs, int w, int h, int uw, int uh, int bitDepth)
{
  int T, L; orc_intra_ref_lengths(w, h, &T, &L);
  const int aboveUnits = (T + uw - 1) / uw, leftUnits = (L + uh - 1) / uh, total = aboveUnits + leftUnits + 1;
  const Pel dc = (Pel)(1 << (bitDepth - 1));
  int numAvail = 0, first = -1;
  for (int u = 0; u < total; u++) if (flags[u]) { numAvail++; if (first < 0) first = u; }
  // recon sample `o` (0 = first in chain order) of unit u
  auto unitSample = [&](int u, int o) -> Pel
  {
    if (u < leftUnits) { const int y = (leftUnits - u) * uh - 1 - o; return rec[(ptrdiff_t)y * recStride - 1]; }      // chain order runs upwards
    if (u == leftUnits) return rec[-recStride - 1];
    return rec[-recStride + (u - leftUnits - 1) * uw + o];
  };
  auto unitLen = [&](int u) { return u < leftUnits ? uh : uw; };
  auto value = [&](int u, int o) -> Pel
  {
    if (numAvail == 0) return dc;
    if (flags[u]) return unitSample(u, o);
    int p = u - 1;
    while (p >= 0 && !flags[p]) p--;
    if (p >= 0) return unitSample(p, unitLen(p) - 1);
    return unitSample(first, 0);
  };
  for (int p = 0; p <= T; p++) { const int q = uw - 1 + p; refs[p] = value(leftUnits + q / uw, q % uw); }
  for (int i = 1; i <= L; i++) { const int k = leftUnits * uh - i; refs[T + i] = value(k / uh, k % uh); }
  return 0;
}
