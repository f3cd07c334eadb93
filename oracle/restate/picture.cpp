// oracle/restate/picture.cpp -- TEST INFRASTRUCTURE: scalar restatement of the picture-level passes of next row N4.
//   Picture::extendPicBorder   CommonLib/Picture.cpp:996-1041
//   compCRC                    CommonLib/PicYuvMD5.cpp:83-125
//   compChecksum               CommonLib/PicYuvMD5.cpp:143-169
// Pinned against the compiled reference (Picture::extendPicBorder on a real Picture, compCRC / compChecksum) by tests/golden/picture.npz.
#include "orc_common.h"

// plane = sample (0,0) inside the padded allocation
ORC_API int orc_extend_border(Pel* plane, int stride, int w, int h, int mx, int my)
{
  for (int y = 0; y < h; y++)
    for (int x = 0; x < mx; x++)
    {
      plane[(ptrdiff_t)y * stride - mx + x] = plane[(ptrdiff_t)y * stride];
      plane[(ptrdiff_t)y * stride + w + x] = plane[(ptrdiff_t)y * stride + w - 1];
    }
  for (int y = 1; y <= my; y++)
  {
    memcpy(plane + (ptrdiff_t)(h - 1 + y) * stride - mx, plane + (ptrdiff_t)(h - 1) * stride - mx, sizeof(Pel) * (w + 2 * mx));
    memcpy(plane - (ptrdiff_t)y * stride - mx, plane - mx, sizeof(Pel) * (w + 2 * mx));
  }
  return 0;
}

ORC_API uint32_t orc_crc(int bitdepth, const Pel* plane, int stride, int w, int h)
{
  uint32_t crc = 0xffff;
  auto feed = [&](unsigned byte) {
    for (int b = 7; b >= 0; b--)
    {
      const uint32_t msb = (crc >> 15) & 1;
      crc = (((crc << 1) + ((byte >> b) & 1)) & 0xffff) ^ (msb * 0x1021);
    }
  };
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
    {
      const int pel = plane[(ptrdiff_t)y * stride + x];
      feed(pel & 0xff);
      if (bitdepth > 8) feed((pel >> 8) & 0xff);
    }
  feed(0); feed(0);
  return crc;
}

ORC_API uint32_t orc_checksum(int bitdepth, const Pel* plane, int stride, int w, int h)
{
  uint32_t sum = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
    {
      const uint8_t mask = (uint8_t)((x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8));
      const int pel = plane[(ptrdiff_t)y * stride + x];
      sum += (uint32_t)((pel & 0xff) ^ mask);
      if (bitdepth > 8) sum += (uint32_t)((pel >> 8) ^ mask);
    }
  return sum;
}
