"""Device-backed mirrors of sygnals.core.{dsp,filters,features}: same function names,
argument meaning, return dtypes and error behaviour as the reference modules."""
