// Policy forward ABI -- placeholder until the fused kernel lands (returns UNSUPPORTED loudly).
struct vnl_policy {
  vnl_policy_spec spec;
  int max_batch, device;
};
extern "C" int vnl_policy_create(const vnl_policy_spec* spec, int32_t max_batch, int32_t device, vnl_policy** out) {
  if (!spec || !out) return fail(VNL_ERR_ARG, "vnl_policy_create: null argument");
  return fail(VNL_ERR_UNSUPPORTED, "vnl_policy_*: not implemented yet");
}
extern "C" void vnl_policy_destroy(vnl_policy* p) { delete p; }
extern "C" int64_t vnl_policy_num_params(const vnl_policy*) { return -1; }
extern "C" int vnl_policy_forward(vnl_policy*, const float*, const float*, const float*, const float*, const float*,
                                  const float*, const float*, int32_t, int32_t, float*, float*, float*, float*,
                                  float*, float*, void*) {
  return fail(VNL_ERR_UNSUPPORTED, "vnl_policy_forward: not implemented yet");
}
