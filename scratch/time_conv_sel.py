import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["SHAPES"] = "4"
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_conv.py")).read())
