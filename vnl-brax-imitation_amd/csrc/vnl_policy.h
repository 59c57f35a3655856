// Intention-network forward (device side declarations).  See vnl_policy_impl.h.
#pragma once
