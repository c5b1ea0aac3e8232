// force-included by tests/test_adapter_syntax.py::test_the_syntax_check_can_fail only
inline int break_on_purpose() { return mugiq_hip_no_such_entry_point(); }
