"""CPU guard of the attitude precise mode's arithmetic ("Which arithmetic has to be double").

The float64 emulation of tests/studies/precise_mode_study.py reproduces the device's open-loop drift to two digits (3.2e-5 measured on
the GPU for the rejected first attempt, 3.2e-5 here).  This test pins the DECISION to it: the shipped split -- float32 sin / cos,
Euler-rate kinematics and acceleration, with T/m and g entering the vertical channel as double-float pairs; double torques, rates and
accumulation -- holds the north-star's 1e-5 over 1000 steps on both scenarios, and the two tempting simplifications do not."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'studies'))
import precise_mode_study as study  # noqa: E402


def test_shipped_attitude_arithmetic_holds_the_drift_bound_and_the_rejected_ones_do_not():
    shipped = dict(trig32=True, kin32=True, acc_df=True)
    hover, circle = study.run_arithmetic('hover', **shipped), study.run_arithmetic('c4', **shipped)
    assert hover[1000] <= 5e-6 and circle[1000] <= 5e-6, (hover, circle)
    # the first attempt of round 3: float32 acceleration products WITHOUT the double-float vertical channel -- fine on the hover
    # scenario, a constant 6e-7 m/s^2 in the vertical acceleration on the circle (what the C4 drift gate on the device caught)
    first = dict(trig32=True, kin32=True, acc32=True)
    assert study.run_arithmetic('hover', steps=300, **first)[300] <= 1e-5
    assert study.run_arithmetic('c4', **first)[1000] > 1e-5
    # thrusts rounded to float32 before the torque differences: the cancellation makes a constant torque error
    assert study.run_arithmetic('hover', steps=300, torque32=True)[300] > 1e-5
