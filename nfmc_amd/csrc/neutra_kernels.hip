// K5: NeuTra HMC (latent-space HMC through the flow).  Placeholder until the VJP kernels land.
#include "flow_device.hpp"

extern "C" int nfmc_neutra_hmc_steps_f32(const NfmcNeutraHmcArgs* args, nfmc_stream_t stream) {
    (void)args; (void)stream;
    return NFMC_EUNSUPPORTED;
}

extern "C" int nfmc_neutra_potential_grad_f32(const NfmcRealNVP* flow, const NfmcPotential* pot, const float* z,
                                              int64_t n, float* u_out, float* grad_out, nfmc_stream_t stream) {
    (void)flow; (void)pot; (void)z; (void)n; (void)u_out; (void)grad_out; (void)stream;
    return NFMC_EUNSUPPORTED;
}
