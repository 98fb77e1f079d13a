# rendering is never called by the golden generator
