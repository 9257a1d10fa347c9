"""Host-side samplers over the C ABI: `mcmc` (MALA / ULA / HMC / UHMC / MH), `jump` (JumpNFMC family), `imh`
(FixedIMH / AdaptiveIMH), `neutra` (NeuTra HMC / MH); `common` holds the per-call device state (`Run`)."""
