"""MI355X-native Omni-Video 2 sampling path: drop-in for the reference's `omnivideo`
package surface on the DiT/VAE/sampler hot path, backed by hand-written HIP kernels
behind a C-ABI library (csrc/ -> libomnihip.so)."""
