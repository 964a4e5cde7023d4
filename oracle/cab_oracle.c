/* cab_oracle.c -- TEST INFRASTRUCTURE (the CPU oracle; see oracle.h).  CPU restatement of the CFDATA checksum,
 * the reference's cabd_checksum() (libmspack/mspack/cabd.c:1462-1479): XOR of the data's little-endian 32-bit
 * words, then the 1-3 trailing bytes folded in BIG-endian order into one more word (3 bytes: b0<<16 | b1<<8 | b2;
 * 2 bytes: b0<<8 | b1; 1 byte: b0).  A CFDATA block's stored checksum is this over the payload (seed 0), then over the
 * header's cbData / cbUncomp fields with that as the seed (cabd.c:1411-1417).
 * PARITY PINNING: tests/test_oracle_golden.py::test_cab_checksum_on_reference_cabinets -- every CFDATA block with a
 * checksum in the reference's own test cabinets (Microsoft-made files among them) carries what this function computes. */
#include "oracle.h"

uint32_t oracle_cab_checksum(const uint8_t *data, size_t bytes, uint32_t cksum)
{
  size_t words = bytes >> 2, i;
  uint32_t tail = 0;
  for (i = 0; i < words; i++, data += 4)
    cksum ^= (uint32_t) data[0] | ((uint32_t) data[1] << 8) | ((uint32_t) data[2] << 16) | ((uint32_t) data[3] << 24);
  switch (bytes & 3) {
  case 3: tail |= (uint32_t) *data++ << 16;  /* fall through */
  case 2: tail |= (uint32_t) *data++ << 8;   /* fall through */
  case 1: tail |= *data;
  }
  return cksum ^ tail;
}
