// Device side of the deterministic PCM generator (pcm_synth_core.h): fills a whole batch in HBM in seconds, so
// that bench.py and the full-size parity runs work on inputs that are a pure function of (seed, stream, sample)
// -- reproducible anywhere, md5-pinned by tests/test_synth.py -- instead of a torch random stream.
// Not on the encoding path; one thread per sample position, all channels.
#include "mp3mi_host.h"
#include "mp3mi.h"
#include "pcm_synth_core.h"

extern "C" void mp3mi_synth_params(int rate_hz, synth_params *P); // pcm_synth_host.cpp

__global__ void __launch_bounds__(256) k_synth(synth_params P, uint32_t seed, uint32_t stream0, int channels, long n_per_ch,
                                               int16_t *__restrict__ out)
{
    const long n = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_per_ch) return;
    const uint32_t s = blockIdx.y;
    int16_t v[2];
    synth_sample(&P, seed, stream0 + s, channels, n, v);
    int16_t *o = out + ((size_t) s * (size_t) n_per_ch + (size_t) n) * (size_t) channels;
    o[0] = v[0];
    if (channels == 2) o[1] = v[1];
}

extern "C" int mp3mi_synth_pcm_device(int16_t *pcm_dev, int n_streams, long n_per_ch, int channels, int rate_hz,
                                      uint32_t stream0, uint32_t seed)
{
    if (!pcm_dev || n_streams <= 0 || n_streams > 65535 || n_per_ch <= 0 || (channels != 1 && channels != 2)) return MP3MI_ERR_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return MP3MI_ERR_NO_DEVICE;
    synth_params P;
    mp3mi_synth_params(rate_hz, &P);
    hipLaunchKernelGGL(k_synth, dim3((unsigned) ((n_per_ch + 255) / 256), (unsigned) n_streams), dim3(256), 0, 0, P, seed, stream0,
                       channels, n_per_ch, pcm_dev);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return MP3MI_ERR_HIP;
    return MP3MI_OK;
}
