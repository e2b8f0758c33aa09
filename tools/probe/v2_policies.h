// Policies for the bound-finding builds of gemm_bf16_v2_kernel (round 2, profiles/r02_contraction_bounds.txt):
//   tools/build_probe_libs.sh nomfma:-include,tools/probe/v2_policies.h,-DSKF_V2_POLICY=V2NoMfma ...
// The product build never sees this file; its policy is skf::V2Full (skf_kernels.h).
#pragma once
namespace skf {
struct V2NoMfma { static constexpr bool stream_a = true, stream_b = true, mfma = false; };     // ingest-only time
struct V2NoDma { static constexpr bool stream_a = false, stream_b = false, mfma = true; };     // compute-only (prologue tiles)
struct V2NoDmaA { static constexpr bool stream_a = false, stream_b = true, mfma = true; };     // without the relation stream
struct V2NoDmaB { static constexpr bool stream_a = true, stream_b = false, mfma = true; };     // without the G^T stream
struct V2IngestA { static constexpr bool stream_a = true, stream_b = false, mfma = false; };
struct V2IngestB { static constexpr bool stream_a = false, stream_b = true, mfma = false; };
}  // namespace skf
