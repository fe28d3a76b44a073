/*
 * wav.h -- minimal RIFF/WAVE reader and writer for the example tools (own implementation; the
 * reference wraps dr_wav in test/wav.h, which is a third-party download and not present here).
 *
 * Reads PCM 16/24/32-bit integer and 32-bit IEEE float files, any channel count, and down-mixes
 * to mono by averaging the channels like the reference's reader (test/wav.h:69-84).  Integer
 * samples are scaled by 1 / 2^(bits-1), the dr_wav convention the reference's C path uses.
 * Writes mono 32-bit IEEE float.
 */

#ifndef SDFT_EXAMPLES_WAV_H
#define SDFT_EXAMPLES_WAV_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t wav_u32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t wav_u16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

/* returns 1 on success; *data is malloc'ed (free it), *size = frames, *samplerate in Hz */
static int wav_read_mono(const char* path, float** data, size_t* size, size_t* samplerate)
{
  FILE* f = fopen(path, "rb");
  if (!f) return 0;
  unsigned char hdr[12];
  if (fread(hdr, 1, 12, f) != 12 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4)) { fclose(f); return 0; }
  unsigned format = 0, channels = 0, bits = 0, rate = 0;
  unsigned char* payload = NULL;
  uint32_t payload_bytes = 0;
  unsigned char ck[8];
  while (fread(ck, 1, 8, f) == 8)
  {
    const uint32_t len = wav_u32(ck + 4);
    if (!memcmp(ck, "fmt ", 4))
    {
      unsigned char fmt[40] = {0};
      const uint32_t take = len < sizeof(fmt) ? len : (uint32_t)sizeof(fmt);
      if (fread(fmt, 1, take, f) != take) break;
      if (len > take) fseek(f, (long)(len - take), SEEK_CUR);
      format = wav_u16(fmt); channels = wav_u16(fmt + 2); rate = wav_u32(fmt + 4); bits = wav_u16(fmt + 14);
      if (format == 0xFFFE && take >= 26) format = wav_u16(fmt + 24);      /* WAVE_FORMAT_EXTENSIBLE */
    }
    else if (!memcmp(ck, "data", 4))
    {
      payload = (unsigned char*)malloc(len ? len : 1);
      payload_bytes = (uint32_t)fread(payload, 1, len, f);
      break;
    }
    else fseek(f, (long)(len + (len & 1)), SEEK_CUR);
  }
  fclose(f);
  if (!payload || !channels || !(bits == 16 || bits == 24 || bits == 32) || !(format == 1 || format == 3)) { free(payload); return 0; }
  const size_t bps = bits / 8, frames = payload_bytes / (bps * channels);
  float* out = (float*)malloc((frames ? frames : 1) * sizeof(float));
  for (size_t i = 0; i < frames; ++i)
  {
    double acc = 0;
    for (unsigned c = 0; c < channels; ++c)
    {
      const unsigned char* p = payload + (i * channels + c) * bps;
      double v;
      if (format == 3) { float fv; memcpy(&fv, p, 4); v = fv; }
      else if (bits == 16) v = (int16_t)wav_u16(p) / 32768.0;
      else if (bits == 24) { int32_t s = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8; v = s / 8388608.0; }
      else v = (int32_t)wav_u32(p) / 2147483648.0;
      acc += v;
    }
    out[i] = (float)(acc / channels);
  }
  free(payload);
  *data = out; *size = frames; *samplerate = rate;
  return 1;
}

static int wav_write_mono_f32(const char* path, const float* data, size_t size, size_t samplerate)
{
  FILE* f = fopen(path, "wb");
  if (!f) return 0;
  const uint32_t bytes = (uint32_t)(size * 4), rate = (uint32_t)samplerate;
  unsigned char h[44] = {'R','I','F','F', 0,0,0,0, 'W','A','V','E', 'f','m','t',' ', 16,0,0,0, 3,0, 1,0, 0,0,0,0, 0,0,0,0, 4,0, 32,0,
                         'd','a','t','a', 0,0,0,0};
  const uint32_t riff = 36 + bytes, byterate = rate * 4;
  memcpy(h + 4, &riff, 4); memcpy(h + 24, &rate, 4); memcpy(h + 28, &byterate, 4); memcpy(h + 40, &bytes, 4);
  fwrite(h, 1, 44, f);
  fwrite(data, 4, size, f);
  fclose(f);
  return 1;
}

#endif
